"""VGPRs / scratch / static LDS of every kernel in a built library (reads the code-object notes; no GPU needed).
usage: python tools/kernel_resources.py [lib.so] [substring ...]"""
import re, struct, subprocess, sys, tempfile, os

def main():
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith('.so') else os.path.join(os.path.dirname(__file__), '..', 'diff-vit_amd', 'csrc', 'libp2vit_hip.so')
    pats = [a for a in sys.argv[1:] if not a.endswith('.so')]
    so = open(lib, 'rb').read()
    i = so.find(b'__CLANG_OFFLOAD_BUNDLE__')
    n = struct.unpack_from('<Q', so, i + 24)[0]
    off = i + 32
    co = None
    for _ in range(n):
        o, sz, tl = struct.unpack_from('<QQQ', so, off); off += 24
        t = so[off:off + tl].decode(); off += tl
        if 'gfx950' in t:
            co = so[i + o:i + o + sz]
    with tempfile.NamedTemporaryFile(suffix='.co') as f:
        f.write(co); f.flush()
        notes = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', f.name], capture_output=True, text=True).stdout
    rows = []
    for b in notes.split('- .agpr_count')[1:]:
        g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, b).group(1)
        rows.append((g('name'), int(g('vgpr_count')), int(g('private_segment_fixed_size')), int(g('group_segment_fixed_size'))))
    dem = subprocess.run(['c++filt'] + [r[0] for r in rows], capture_output=True, text=True).stdout.split('\n')
    print('vgpr scratch lds  kernel')
    for r, d in zip(rows, dem):
        d = d.replace('void ', '').split('(')[0]
        if not pats or any(p in d for p in pats) or (pats == ['scratch'] and r[2] > 0):
            print('%4d %7d %6d  %s' % (r[1], r[2], r[3], d))

main()
