"""Real-data leg of the evaluation harness: the reference's ``build_transform`` and ImageFolder loaders (test_quant.py:100-144,
504-534) without torchvision (not installed here) -- PIL + torch only.

``build_transform`` = Resize(floor(input/crop_pct), bicubic) -> CenterCrop(input) -> ToTensor -> Normalize with torchvision's
geometry rules (shorter side to ``size``, the other side ``int(size * long / short)``; crop offsets ``int(round((dim - crop) / 2))``).
**Parity unpinned**: nothing in the reference pins the transform's pixels and torchvision cannot be imported here to compare; the
resampling itself is PIL's, which is also what torchvision calls for PIL inputs.  ``ImageFolder`` follows torchvision's convention
(classes = sorted sub-directory names, samples sorted by path, the usual image extensions).
"""
import math
import os

import numpy as np
import torch
from PIL import Image

IMG_EXTENSIONS = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp', '.pgm', '.tif', '.tiff', '.webp')
_INTERP = {'bicubic': Image.BICUBIC, 'lanczos': Image.LANCZOS, 'hamming': Image.HAMMING}

# per model family, test_quant.py:100-113
MODEL_STATS = {'deit': ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225), 0.875),
               'vit': ((0.5, 0.5, 0.5), (0.5, 0.5, 0.5), 0.9),
               'swin': ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225), 0.9)}


def build_transform(input_size=224, interpolation='bicubic', mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), crop_pct=0.875):
    """test_quant.py:504-534: PIL image -> normalised fp32 tensor [3, input_size, input_size]."""
    method = _INTERP.get(interpolation, Image.BILINEAR)
    size = int(math.floor(input_size / crop_pct)) if input_size > 32 else None
    mean_t = torch.tensor(mean, dtype=torch.float32).reshape(3, 1, 1)
    std_t = torch.tensor(std, dtype=torch.float32).reshape(3, 1, 1)

    def transform(img):
        img = img.convert('RGB')
        if size is not None:
            w, h = img.size
            if (w <= h and w != size) or (h <= w and h != size):              # Resize(int): shorter side -> size
                nw, nh = (size, int(size * h / w)) if w <= h else (int(size * w / h), size)
                img = img.resize((nw, nh), method)
            w, h = img.size
            if w < input_size or h < input_size:                                # CenterCrop pads small images with zeros
                pad = Image.new('RGB', (max(w, input_size), max(h, input_size)))
                pad.paste(img, ((pad.size[0] - w) // 2, (pad.size[1] - h) // 2))
                img, (w, h) = pad, pad.size
            top, left = int(round((h - input_size) / 2.0)), int(round((w - input_size) / 2.0))
            img = img.crop((left, top, left + input_size, top + input_size))
        x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)   # ToTensor
        return (x - mean_t) / std_t                                                                        # Normalize

    return transform


class ImageFolder(torch.utils.data.Dataset):
    """root/<class>/<image>: (transformed image, class index), torchvision.datasets.ImageFolder's ordering (test_quant.py:122,136)."""

    def __init__(self, root, transform=None):
        self.root, self.transform = root, transform
        self.classes = sorted(d.name for d in os.scandir(root) if d.is_dir())
        if not self.classes:
            raise FileNotFoundError("Couldn't find any class folder in %s." % root)
        self.class_to_idx = {c: i for i, c in enumerate(self.classes)}
        self.samples = []
        for c in self.classes:
            for dirpath, _, files in sorted(os.walk(os.path.join(root, c), followlinks=True)):
                for f in sorted(files):
                    if f.lower().endswith(IMG_EXTENSIONS):
                        self.samples.append((os.path.join(dirpath, f), self.class_to_idx[c]))
        if not self.samples:
            raise FileNotFoundError('Found no valid file for the classes in %s.' % root)

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        path, target = self.samples[i]
        with Image.open(path) as img:
            x = img.convert('RGB')
        return (self.transform(x) if self.transform else x), target


def build_loaders(data_root, model_name, val_batchsize, calib_batchsize, num_workers=0):
    """the reference's two loaders (test_quant.py:118-144): val (in order) and train (shuffled, drop_last; calibration batches)."""
    family = 'swin' if model_name.startswith('swin') else ('vit' if model_name.startswith('vit') else 'deit')
    mean, std, crop_pct = MODEL_STATS[family]
    tf = build_transform(mean=mean, std=std, crop_pct=crop_pct)
    val = torch.utils.data.DataLoader(ImageFolder(os.path.join(data_root, 'val'), tf), batch_size=val_batchsize, shuffle=False,
                                      num_workers=num_workers, pin_memory=torch.cuda.is_available())       # (test_quant.py:128: pin_memory=True)
    train_dir = os.path.join(data_root, 'train')
    train = None
    if os.path.isdir(train_dir):
        train = torch.utils.data.DataLoader(ImageFolder(train_dir, tf), batch_size=calib_batchsize, shuffle=True, num_workers=num_workers,
                                            pin_memory=torch.cuda.is_available(), drop_last=True)
    return val, train
