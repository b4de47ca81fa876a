"""Build libdynamask_hip.so (gfx950) in-tree with hipcc.  No torch involved:
the device code is HIP from the start, the library is a plain C-ABI .so."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libdynamask_hip.so')
SOURCES = ['api_misc.hip', 'roi_align.hip', 'conv_igemm.hip', 'deform_conv.hip', 'pointwise.hip', 'carafe.hip', 'mask_pre.hip', 'backward.hip', 'rle.hip', 'bbox.hip', 'fc_gemm.hip', 'bbox_train.hip', 'polygon.hip', 'dcn_bwd_fused.hip']


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build_library(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, 'common.h'), os.path.join(ROOT, 'include', 'dynamask_hip.h')]
    if not force and _newer(LIB, deps):
        return LIB
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + '.o')
        if force or not _newer(o, [s, deps[-2], deps[-1]]):
            cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-I' + os.path.join(ROOT, 'include'),
                   '-I' + CSRC, '-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build_library(force='--force' in sys.argv)
