"""CPU-side checks: the C-ABI library loads and exports every symbol the header
declares (no compute), the registry/config boundary, state_dict compatibility,
and that the product refuses to run without a HIP device."""
import os
import re

import pytest
import torch

import golden_inputs as gi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    from dynamask_amd import build
    return build.build_library(verbose=False)


def test_library_exports_every_declared_symbol():
    path = _build()
    assert os.path.exists(path)
    hdr = open(os.path.join(ROOT, 'include', 'dynamask_hip.h')).read()
    declared = set(re.findall(r'\b(dm_[a-z0-9_]+)\s*\(', hdr))
    declared.discard('dm_error_string')
    declared.add('dm_error_string')
    from dynamask_amd import _lib
    L = _lib.lib()
    assert set(_lib.SIGNATURES) == declared, declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(L, name), name
    assert L.dm_abi_version() == _lib.ABI_VERSION
    assert L.dm_conv_packed_cout(36) == 64 and L.dm_conv_packed_cout(256) == 256
    assert L.dm_error_string(-1).decode().startswith('invalid')


def test_library_says_how_it_was_built_and_the_loader_checks_it(tmp_path, monkeypatch):
    """dm_build_info() carries the product-wide flags of build.py; _lib.lib() refuses a library that does not say it was
    built without packed fp32 (a recipe that drops the flag must not load silently)."""
    import subprocess
    from dynamask_amd import _lib, build
    L = _lib.lib()
    info = L.dm_build_info().decode()
    assert f'abi={_lib.ABI_VERSION}' in info and 'gfx950' in info and 'clang' in info.lower()
    for f in build.FLAGS:
        assert f in info, (f, info)
    assert _lib.REQUIRED_BUILD_FLAG in info
    # a library compiled by "some other recipe": the one translation unit that holds dm_build_info, no flag handed over
    src = os.path.join(ROOT, 'dynamask_amd', 'csrc', 'api_misc.hip')
    other = tmp_path / 'libother.so'
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O1', '-fPIC', '-std=c++17', '-shared',
                           '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.dirname(src), src, '-o', str(other)])
    import ctypes
    O = ctypes.CDLL(str(other))
    O.dm_build_info.restype = ctypes.c_char_p
    assert O.dm_build_info().decode().endswith('flags=unknown')
    with pytest.raises(_lib.DynaMaskLibraryError, match='packed fp32'):
        _lib.check_build_info(O.dm_build_info().decode())
    monkeypatch.setenv('DM_ALLOW_PACKED_FP32', '1')
    _lib.check_build_info(O.dm_build_info().decode())          # the A/B escape hatch
    monkeypatch.delenv('DM_ALLOW_PACKED_FP32')
    _lib.check_build_info(info)


def test_argument_validation_without_gpu():
    from dynamask_amd import _lib
    L = _lib.lib()
    # null pointers / bad sizes are rejected before any HIP call
    assert L.dm_upsample2x_bilinear_fwd(None, 1, 4, 4, 0, 0, None, None) == -1
    assert L.dm_conv_pack_weight(None, 4, 4, 3, 0, 1, None, None, None) == -1
    assert L.dm_gumbel_select_fwd(None, None, 4, 4, 0.5, None, None, None, None) == -1


def test_ops_fail_loudly_on_cpu_tensors():
    from dynamask_amd import ops
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.upsample2x(torch.zeros(1, 1, 4, 4))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.roi_align([torch.zeros(1, 1, 4, 4)], torch.zeros(1, 5), 7, [1.0])


def _cfg():
    from dynamask_amd import registry
    return dict(type='DynaMaskRoIHead',
                mask_roi_extractor=dict(type='SingleRoIExtractor', **gi.MASK_ROI_EXTRACTOR_CFG),
                mask_head=dict(type='DynaMaskHead', **gi.MASK_HEAD_CFG),
                train_cfg=registry.ConfigDict(flops=[0.23, 0.62, 1.01, 1.4], Lambda=0.3, mask_size=28),
                test_cfg=registry.ConfigDict(mask_thr_binary=0.5))


def test_registry_builds_roi_head_with_reference_state_dict_keys():
    from dynamask_amd import registry, roi_head, losses, mask_heads, roi_extractors  # noqa: F401
    m = registry.build_head(_cfg())
    ref = {**gi.head_state(), **gi.mask_pre_state()}
    assert set(m.state_dict()) == set(ref)
    m.load_state_dict(ref, strict=True)
    n_head = sum(v.numel() for k, v in m.state_dict().items() if k.startswith('mask_head.'))
    n_pre = sum(p.numel() for p in m.mask_predictor.parameters())
    assert n_head == 2502632 + 2 and n_pre == 1659828            # flat gradient buffer = 4 162 462 floats
    for name in ('DynaMaskRoIHead', 'DynaMaskHead', 'FCNMaskHead'):
        assert name in registry.HEADS
    assert 'SingleRoIExtractor' in registry.ROI_EXTRACTORS and 'DynaCrossEntropyLoss' in registry.LOSSES
    with pytest.raises(KeyError):
        registry.build_head(dict(type='NoSuchHead'))


@pytest.mark.skipif(not os.path.exists('/root/reference/configs/dynamask/coco/r50-dynamask-1x.py'),
                    reason='reference tree only exists in the authoring container')
def test_reference_config_is_consumed_unchanged():
    from dynamask_amd import registry, roi_head, losses, mask_heads, roi_extractors  # noqa: F401
    cfg = registry.Config.fromfile('/root/reference/configs/dynamask/coco/r50-dynamask-1x.py')
    rh = dict(cfg.model.roi_head)
    rh.update(train_cfg=cfg.train_cfg.rcnn, test_cfg=cfg.test_cfg.rcnn)
    m = registry.build_head(rh)
    assert m.train_cfg.flops == [0.23, 0.62, 1.01, 1.4] and m.test_cfg.mask_thr_binary == 0.5
    assert m.mask_head.loss_func.start_stage == 4 and m.mask_head.loss_func.cb_loss_weight == 0.8
    assert m.mask_roi_extractor.featmap_strides == [4, 8, 16, 32]
    # bbox branch (8f rank 4): built from the same unchanged config, reference state_dict keys
    assert m.with_bbox and m.bbox_head.num_classes == 80 and m.bbox_head.bbox_coder.stds == (0.1, 0.1, 0.2, 0.2)
    keys = set(m.state_dict().keys())
    for k in ('shared_fcs.0.weight', 'shared_fcs.1.bias', 'fc_cls.weight', 'fc_reg.bias'):
        assert 'bbox_head.' + k in keys
    assert tuple(m.bbox_head.shared_fcs[0].weight.shape) == (1024, 12544) and tuple(m.bbox_head.fc_reg.weight.shape) == (320, 1024)
    # the values restated in tests/golden/golden_inputs.py are the config's
    assert dict(cfg.model.roi_head.mask_head.loss_cfg) == dict(type='DynaCrossEntropyLoss', **gi.LOSS_CFG)


def test_reference_carafe_config_builds_the_standard_roi_head_unchanged():
    """configs/carafe/mask_rcnn_r50_fpn_carafe_1x_coco.py (BASELINE configs[4]: StandardRoIHead + FCNMaskHead with the
    CARAFE upsample, two ``_base_`` levels deep) is consumed unchanged; the module tree has the reference's state_dict
    keys, incl. the ``mask_predictor.*`` block the fork's BaseRoIHead gives every RoI head (Quirk Q4)."""
    from dynamask_amd import registry, roi_head, losses, mask_heads, roi_extractors, bbox_heads  # noqa: F401
    cfg = registry.Config.fromfile('/root/reference/configs/carafe/mask_rcnn_r50_fpn_carafe_1x_coco.py')
    rh = dict(cfg.model.roi_head)
    assert rh['type'] == 'StandardRoIHead' and rh['mask_head']['type'] == 'FCNMaskHead'
    rh.update(train_cfg=cfg.train_cfg.rcnn, test_cfg=cfg.test_cfg.rcnn)
    m = registry.build_head(rh)
    assert type(m).__name__ == 'StandardRoIHead' and m.mask_head.upsample_method == 'carafe' and m.mask_head.num_classes == 80
    assert m.train_cfg.mask_size == 28 and m.test_cfg.mask_thr_binary == 0.5 and m.with_bbox and m.with_mask
    keys = set(m.state_dict())
    ref = set(gi.fcn_state('carafe')) | {'bbox_head.' + k for k in ('shared_fcs.0.weight', 'shared_fcs.0.bias', 'shared_fcs.1.weight',
                                                                   'shared_fcs.1.bias', 'fc_cls.weight', 'fc_cls.bias', 'fc_reg.weight', 'fc_reg.bias')}
    assert ref <= keys
    assert {k for k in keys if not (k in ref)} == {k for k in keys if k.startswith('mask_predictor.')} and 'mask_predictor.fc2.weight' in keys


def test_fcn_head_state_dict_keys():
    from dynamask_amd import registry, mask_heads  # noqa: F401
    for up in ('deconv', 'carafe'):
        cfg = dict(type='FCNMaskHead', num_convs=4, in_channels=256, conv_out_channels=256, num_classes=80)
        if up == 'carafe':
            cfg['upsample_cfg'] = dict(type='carafe', scale_factor=2, up_kernel=5, up_group=1, encoder_kernel=3,
                                       encoder_dilation=1, compressed_channels=64)
        h = registry.build_head(cfg)
        ref = {k[len('mask_head.'):]: v for k, v in gi.fcn_state(up).items()}
        assert set(h.state_dict()) == set(ref)
        h.load_state_dict(ref, strict=True)
    with pytest.raises(ValueError):
        registry.build_head(dict(type='FCNMaskHead', upsample_cfg=dict(type='bogus', scale_factor=2)))


def test_library_contains_no_packed_fp32_instructions(tmp_path):
    """Round 5: a compiler-generated ``v_pk_fma_f32 ... op_sel:[0,1,0]`` dropped a product in lane 48 when four queues
    shared the CUs (dynamask_amd/build.py ``FLAGS``; profiles/r05_race_hunt.txt).  The library is built without packed
    fp32 operations; this disassembles the gfx950 code objects of the built .so and fails on any of them."""
    import glob
    import shutil
    import subprocess
    from dynamask_amd import _lib
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    if not os.path.exists(objdump) or not os.path.exists(_lib.LIB_PATH):
        pytest.skip('llvm-objdump or the built library is not here')
    so = shutil.copy(_lib.LIB_PATH, tmp_path / 'lib.so')
    subprocess.run([objdump, '--offloading', str(so)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp_path)
    objs = glob.glob(str(tmp_path / 'lib.so.*gfx950*'))
    assert objs, 'no gfx950 code object found in the library'
    bad = 0
    total = 0
    for o in objs:
        out = subprocess.run([objdump, '-d', o], check=True, capture_output=True, text=True).stdout
        total += out.count('v_mfma_f32')
        bad += sum(out.count(op) for op in ('v_pk_fma_f32', 'v_pk_mul_f32', 'v_pk_add_f32'))
    assert total > 1000, 'the disassembly does not look like the operator library'
    assert bad == 0, f'{bad} packed fp32 instructions in the library: build with dynamask_amd.build.FLAGS'


def test_graph_weight_key_sees_replaced_parameters_and_modules():
    """graphs.GraphedMaskLogits._weights_key: the cached walk must notice a Parameter OBJECT that was
    replaced, not only in-place updates -- otherwise a HIP graph keeps replaying the old weights."""
    import torch.nn as nn
    from dynamask_amd import graphs, registry
    head = registry.build_head(_cfg())
    g = graphs.GraphedMaskLogits(head)
    k0 = g._weights_key()
    assert g._weights_key() == k0
    mh = head.mask_head
    with torch.no_grad():
        mh.stages[0].instance_logits.weight.add_(1.0)                          # in place: version
    k1 = g._weights_key()
    assert k1 != k0
    mh.stages[1].fuse_transform_out.weight = nn.Parameter(mh.stages[1].fuse_transform_out.weight.detach().clone())
    k2 = g._weights_key()
    assert k2 != k1                                                             # replaced object: address
    sd = {k: v.clone() for k, v in mh.state_dict().items()}
    mh.load_state_dict(sd, assign=True)
    k3 = g._weights_key()
    assert k3 != k2
    from dynamask_amd import mask_heads
    mh.instance_convs[1] = mask_heads.ConvModule(256, 256, 3, padding=1)       # replaced submodule
    k4 = g._weights_key()
    assert k4 != k3 and len(k4[0]) == len(k0[0])
    mh.stages[2].fuse_conv[0].bias = None                                       # a slot that disappears
    assert len(g._weights_key()[0]) == len(k0[0]) - 1
