"""VGPRs / scratch / static LDS of every kernel in a built library, read from the code-object notes (llvm-readelf --notes on the unbundled
gfx950 code objects; no GPU needed).  The library holds one offload bundle per translation unit.
usage: python tools/kernel_resources.py [lib.so] [substring ...]          table (substring `scratch`: only kernels with a private segment)
       python tools/kernel_resources.py [lib.so] --isa <substring> [out]   disassembly of the first matching kernel + instruction-class counts"""
import collections, os, re, struct, subprocess, sys, tempfile

LLVM = '/opt/rocm/lib/llvm/bin/'


def code_objects(lib):
    so = open(lib, 'rb').read()
    out, i = [], 0
    while True:
        i = so.find(b'__CLANG_OFFLOAD_BUNDLE__', i)
        if i < 0:
            return out
        n = struct.unpack_from('<Q', so, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from('<QQQ', so, off); off += 24
            t = so[off:off + tl].decode(); off += tl
            if 'gfx950' in t and sz:
                out.append(so[i + o:i + o + sz])
        i += 24


def rows_of(co):
    with tempfile.NamedTemporaryFile(suffix='.co') as f:
        f.write(co); f.flush()
        notes = subprocess.run([LLVM + 'llvm-readelf', '--notes', f.name], capture_output=True, text=True).stdout
    rows = []
    for b in notes.split('- .agpr_count')[1:]:
        g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, b).group(1)
        rows.append((g('name'), int(g('vgpr_count')), int(g('private_segment_fixed_size')), int(g('group_segment_fixed_size'))))
    return rows


def demangle(names):
    return subprocess.run(['c++filt'] + list(names), capture_output=True, text=True).stdout.split('\n')


def isa(cos, pat, out_path):
    for co in cos:
        rows = rows_of(co)
        for r, d in zip(rows, demangle(r[0] for r in rows)):
            if pat in d:
                with tempfile.NamedTemporaryFile(suffix='.co') as f:
                    f.write(co); f.flush()
                    full = subprocess.run([LLVM + 'llvm-objdump', '-d', f.name], capture_output=True, text=True).stdout
                a = full.index('<%s>:' % r[0])
                m = re.search(r'^[0-9a-f]+ <[^>]+>:', full[a + 10:], re.M)
                txt = full[a:a + 10 + m.start()] if m else full[a:]
                ins = [l.split()[0] for l in txt.split('\n') if re.match(r'^\s+[vs]_|^\s+(ds|global|buffer|scratch|flat)_', l)]
                cls = collections.Counter('mfma' if 'mfma' in i else i.split('_')[0] if i[0] in 'vs' else i.split('_')[0] for i in ins)
                print('%s\n  vgpr %d scratch %d lds %d; %d instructions: %s' % (d.split('(')[0], r[1], r[2], r[3], len(ins), dict(cls)))
                print('  most frequent:', collections.Counter(ins).most_common(24))
                if out_path:
                    open(out_path, 'w').write(txt)
                return
    print('no kernel matches', pat)


def main():
    args = sys.argv[1:]
    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'diff-vit_amd', 'csrc', 'libp2vit_hip.so')
    if args and args[0].endswith('.so'):
        lib = args.pop(0)
    cos = code_objects(lib)
    if args and args[0] == '--isa':
        return isa(cos, args[1], args[2] if len(args) > 2 else None)
    print('vgpr scratch lds  kernel')
    for co in cos:
        rows = rows_of(co)
        for r, d in zip(rows, demangle(r[0] for r in rows)):
            d = d.replace('void ', '').split('(')[0]
            if not args or any(p in d for p in args) or (args == ['scratch'] and r[2] > 0):
                print('%4d %7d %6d  %s' % (r[1], r[2], r[3], d))


main()
