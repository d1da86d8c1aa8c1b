"""Builds object_detection_amd/libdetbox.so from csrc/*.hip with hipcc for gfx950.

    python -m object_detection_amd.build [--force]

-ffp-contract=off is part of the numerical contract (no FMA contraction: IoU, box
conversion and encode/decode round exactly like the fp32 reference arithmetic).
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libdetbox.so')
OBJ = os.path.join(HERE, 'csrc', '_obj')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wall', '-Wno-unused-function']


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def source_hash():
    """sha256 over the kernel sources and the C header: identifies the code a counter pass was taken on."""
    import hashlib
    h = hashlib.sha256()
    for path in sources() + sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [os.path.join(HERE, '..', 'include', 'detbox.h')]:
        with open(path, 'rb') as f:
            h.update(os.path.basename(path).encode() + b'\0' + f.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(OUT):
        return True
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(HERE, '..', 'include', 'detbox.h')]
    return any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(OBJ, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(OBJ, os.path.basename(src) + '.o')
        cmd = [hipcc] + FLAGS + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(sources()))) as ex:
        objs = list(ex.map(compile_one, sources()))
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
