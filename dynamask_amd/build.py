"""Build libdynamask_hip.so (gfx950) in-tree with hipcc.  No torch involved:
the device code is HIP from the start, the library is a plain C-ABI .so."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libdynamask_hip.so')
SOURCES = ['api_misc.hip', 'roi_align.hip', 'conv_igemm.hip', 'deform_conv.hip', 'pointwise.hip', 'carafe.hip', 'mask_pre.hip', 'backward.hip', 'rle.hip', 'bbox.hip', 'fc_gemm.hip', 'bbox_train.hip', 'polygon.hip']


# Product-wide compile flags (beside -O3 -fPIC -std=c++17 --offload-arch=gfx950).
# NO packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32), round 5: the compiler pairs neighbouring scalar fp32
# operations into these, and in the training step -- four queues sharing the CUs -- a `v_pk_fma_f32 ... op_sel:[0,1,0]` it
# had generated for class_logits_bwd_wave_kernel returned, in 22 of 2.6 M wave-iterations, a LOW half without its product
# (= the addend), always in lane 48, while two scalar v_fma_f32 on the same registers in the same wave were right
# (profiles/r05_race_hunt.txt: the self-check ran inside the kernel).  That was the "gradient that depends on what runs
# beside it" of round 4.  It never showed with the kernel alone on the GPU, so no test of a kernel by itself can clear
# the other 3 991 packed instructions of the library: they are not generated any more (headline 278.1 -> 277.2 img/s,
# training step 20.10 -> 20.24 ms, RoIAlign 14x14 51.1 -> 52.4 us: inside the run-to-run noise except the last).
# tests/test_host_cpu.py disassembles the built library and fails on any packed fp32 instruction.  (The flag also reaches
# the host pass of hipcc, which prints "'-packed-fp32-ops' is not a recognized feature for this target": harmless.)
NO_PACKED_FP32 = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
FLAGS = list(NO_PACKED_FP32)


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build_library(force=False, verbose=True, extra_flags=(), lib=LIB, objdir=None):
    """``extra_flags`` / ``lib`` / ``objdir``: a second build beside the product's (A/B measurements: tools/)."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, 'common.h'), os.path.join(ROOT, 'include', 'dynamask_hip.h'), os.path.abspath(__file__)]
    if not force and _newer(lib, deps):
        return lib
    objdir = objdir or os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + '.o')
        if force or not _newer(o, [s] + deps[-3:]):
            # the flag set travels into the library (dm_build_info): _lib.py refuses one that lacks -packed-fp32-ops
            said = ' '.join([*FLAGS, *extra_flags]) or 'none'
            cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', *FLAGS, *extra_flags, f'-DDM_BUILD_FLAGS="{said}"',
                   '-I' + os.path.join(ROOT, 'include'),
                   '-I' + CSRC, '-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == '__main__':
    if '--packed-fp32' in sys.argv:        # an A/B build WITH the packed instructions: libdynamask_hip_pk.so beside the product's
        FLAGS.clear()
        build_library(force='--force' in sys.argv, lib=os.path.join(HERE, 'libdynamask_hip_pk.so'), objdir=os.path.join(HERE, 'build', 'pk'))
    else:
        build_library(force='--force' in sys.argv)
