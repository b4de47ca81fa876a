"""Oracle: CPU (PyTorch fp32) restatement of the operators the hot path calls.

TEST INFRASTRUCTURE ONLY -- never imported by the product (dynamask_amd/).

The reference takes these operators from ``mmcv-full==1.0.5`` (pin:
/root/reference/mmdet/__init__.py:17-18), which is NOT in /root/reference and
not installed here, so RoIAlign / SimpleRoIAlign / CARAFE are restated from
mmcv's published algorithm ("parity unpinned").
DCNv1 follows the in-tree spec mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu.
Reference call sites are cited per function.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# RoIAlign (mmcv.ops.roi_align, pool_mode='avg')
#   call sites: mmdet/models/roi_heads/roi_extractors/base_roi_extractor.py:49-55
#               (RoIAlign(output_size, sampling_ratio=0), spatial_scale=1/stride,
#                aligned=True default),
#               mmdet/models/roi_heads/base_roi_head.py:53-57 (56x56 on P2)
# --------------------------------------------------------------------------
def _axis_samples(start, bin_size, grid, pooled, size):
    """Per-axis sample coordinates of one RoI -> (low, high, w_low, w_high, valid).

    Sample s = p*grid + i sits at  start + p*bin + (i+.5)*bin/grid  (fp32).
    Rules (mmcv bilinear_interpolate): coordinate < -1 or > size -> sample is
    void; clamp to >= 0; low = int(c); if low >= size-1: low = high = size-1,
    c = low; weights  w_high = c - low,  w_low = 1 - w_high.
    """
    f32 = torch.float32
    p = torch.arange(pooled, dtype=f32).repeat_interleave(grid)
    i = torch.arange(grid, dtype=f32).repeat(pooled)
    start = torch.tensor(start, dtype=f32)
    bin_size = torch.tensor(bin_size, dtype=f32)
    c = start + p * bin_size + (i + 0.5) * bin_size / torch.tensor(float(grid), dtype=f32)
    valid = ~((c < -1.0) | (c > float(size)))
    c = torch.clamp(c, min=0.0)
    low = c.to(torch.int64)
    edge = low >= size - 1
    high = torch.where(edge, torch.full_like(low, size - 1), low + 1)
    low = torch.where(edge, torch.full_like(low, size - 1), low)
    c = torch.where(edge, low.to(f32), c)
    w_high = c - low.to(f32)
    w_low = 1.0 - w_high
    # void samples contribute 0; park their indices in range
    low = torch.where(valid, low, torch.zeros_like(low))
    high = torch.where(valid, high, torch.zeros_like(high))
    return low, high, w_low, w_high, valid


def roi_align(feat, rois, output_size, spatial_scale=1.0, sampling_ratio=0, aligned=True):
    """feat [B,C,H,W] fp32, rois [K,5]=(batch_idx,x1,y1,x2,y2) -> [K,C,P,P]."""
    if isinstance(output_size, int):
        ph_n = pw_n = output_size
    else:
        ph_n, pw_n = output_size
    if feat.dtype != torch.float64:          # (float64 maps: the fp64 triangle of the gradient tests -- geometry stays fp32)
        feat = feat.float()
    rois = rois.float()
    B, C, H, W = feat.shape
    K = rois.shape[0]
    out = feat.new_zeros((K, C, ph_n, pw_n))
    offset = 0.5 if aligned else 0.0
    f32 = torch.float32
    scale = torch.tensor(spatial_scale, dtype=f32)
    for k in range(K):
        b = int(rois[k, 0].item())
        sw = float(rois[k, 1] * scale - offset)
        sh = float(rois[k, 2] * scale - offset)
        ew = float(rois[k, 3] * scale - offset)
        eh = float(rois[k, 4] * scale - offset)
        rw = float(torch.tensor(ew, dtype=f32) - torch.tensor(sw, dtype=f32))
        rh = float(torch.tensor(eh, dtype=f32) - torch.tensor(sh, dtype=f32))
        if not aligned:
            rw = max(rw, 1.0)
            rh = max(rh, 1.0)
        bin_h = float(torch.tensor(rh, dtype=f32) / torch.tensor(float(ph_n), dtype=f32))
        bin_w = float(torch.tensor(rw, dtype=f32) / torch.tensor(float(pw_n), dtype=f32))
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(
            float(torch.tensor(rh, dtype=f32) / torch.tensor(float(ph_n), dtype=f32))))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(
            float(torch.tensor(rw, dtype=f32) / torch.tensor(float(pw_n), dtype=f32))))
        count = max(gh * gw, 1)
        if gh <= 0 or gw <= 0:
            continue  # empty sample grid -> zeros (loops do not run in mmcv)
        yl, yh, wyl, wyh, vy = _axis_samples(sh, bin_h, gh, ph_n, H)
        xl, xh, wxl, wxh, vx = _axis_samples(sw, bin_w, gw, pw_n, W)
        wyl = (wyl * vy).to(feat.dtype)
        wyh = (wyh * vy).to(feat.dtype)
        wxl = (wxl * vx).to(feat.dtype)
        wxh = (wxh * vx).to(feat.dtype)
        f = feat[b]  # [C,H,W]
        top = f[:, yl, :]
        bot = f[:, yh, :]
        v = (wyl[None, :, None] * wxl[None, None, :]) * top[:, :, xl] \
            + (wyl[None, :, None] * wxh[None, None, :]) * top[:, :, xh] \
            + (wyh[None, :, None] * wxl[None, None, :]) * bot[:, :, xl] \
            + (wyh[None, :, None] * wxh[None, None, :]) * bot[:, :, xh]
        v = v.view(C, ph_n, gh, pw_n, gw).sum(dim=(2, 4)) / float(count)
        out[k] = v
    return out


def roi_align_bruteforce_f64(feat, rois, output_size, spatial_scale=1.0, sampling_ratio=0, aligned=True):
    """Sample-by-sample float64 loop restatement (tiny inputs only): the
    independent cross-check of ``roi_align``."""
    P = output_size
    feat = feat.double()
    B, C, H, W = feat.shape
    K = rois.shape[0]
    out = torch.zeros((K, C, P, P), dtype=torch.float64)
    off = 0.5 if aligned else 0.0

    def interp(f, y, x):
        if y < -1.0 or y > H or x < -1.0 or x > W:
            return torch.zeros(C, dtype=torch.float64)
        y = max(y, 0.0)
        x = max(x, 0.0)
        yl, xl = int(y), int(x)
        if yl >= H - 1:
            yh = yl = H - 1
            y = float(yl)
        else:
            yh = yl + 1
        if xl >= W - 1:
            xh = xl = W - 1
            x = float(xl)
        else:
            xh = xl + 1
        ly, lx = y - yl, x - xl
        hy, hx = 1.0 - ly, 1.0 - lx
        return hy * hx * f[:, yl, xl] + hy * lx * f[:, yl, xh] + ly * hx * f[:, yh, xl] + ly * lx * f[:, yh, xh]

    for k in range(K):
        b = int(rois[k, 0])
        sw, sh, ew, eh = [float(rois[k, j]) * spatial_scale - off for j in (1, 2, 3, 4)]
        rw, rh = ew - sw, eh - sh
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        bh, bw = rh / P, rw / P
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / P))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / P))
        cnt = max(gh * gw, 1)
        for ph in range(P):
            for pw in range(P):
                acc = torch.zeros(C, dtype=torch.float64)
                for iy in range(gh):
                    y = sh + ph * bh + (iy + 0.5) * bh / gh
                    for ix in range(gw):
                        x = sw + pw * bw + (ix + 0.5) * bw / gw
                        acc += interp(feat[b], y, x)
                out[k, :, ph, pw] = acc / cnt
    return out


def map_roi_levels(rois, num_levels, finest_scale=56):
    """FPN level of each RoI.  Follows
    roi_extractors/single_level_roi_extractor.py:32-51."""
    scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    lvls = torch.floor(torch.log2(scale / finest_scale + 1e-6))
    return lvls.clamp(min=0, max=num_levels - 1).long()


def single_roi_extractor(feats, rois, output_size, featmap_strides, finest_scale=56, sampling_ratio=0):
    """SingleRoIExtractor.forward (single_level_roi_extractor.py:53-81)."""
    C = feats[0].shape[1]
    out = feats[0].new_zeros((rois.shape[0], C, output_size, output_size))
    if len(feats) == 1:
        if rois.shape[0] == 0:
            return out
        return roi_align(feats[0], rois, output_size, 1.0 / featmap_strides[0], sampling_ratio, True)
    lvls = map_roi_levels(rois, len(feats), finest_scale)
    for i in range(len(feats)):
        inds = lvls == i
        if inds.any():
            out[inds] = roi_align(feats[i], rois[inds], output_size, 1.0 / featmap_strides[i], sampling_ratio, True)
    return out


# --------------------------------------------------------------------------
# SimpleRoIAlign / point_sample (mmcv/ops/point_sample.py)
#   call site: mask_heads/dynamask_head.py:74,105
# --------------------------------------------------------------------------
def simple_roi_align(feat, rois, output_size, spatial_scale, aligned=True):
    """feat [B,C,H,W]; rois [K,5] grouped by image index -> [K,C,S,S].

    grid = pixel centres (i+.5)/S in RoI-relative [0,1]^2; absolute image
    point = rel*(x2-x1, y2-y1)+(x1,y1); relative-to-feature-map point =
    abs / (W_feat, H_feat) * spatial_scale; sampled with
    F.grid_sample(bilinear, zeros, align_corners=not aligned).
    The per-image outputs are concatenated in image order (mmcv does
    ``torch.cat(point_feats)``), so RoIs must be sorted by batch index,
    which bbox2roi (mmdet/core/bbox/transforms.py:54-73) guarantees.
    """
    S = output_size
    B, C, H, W = feat.shape
    K = rois.shape[0]
    theta = torch.tensor([[[1., 0., 0.], [0., 1., 0.]]], dtype=torch.float32)
    grid = F.affine_grid(theta, torch.Size((1, 1, S, S)), align_corners=False)
    grid = (grid + 1.0) / 2.0                      # [-1,1] -> [0,1]
    rel = grid.view(1, -1, 2).expand(K, -1, -1)     # [K, S*S, 2] (x, y)
    outs = []
    for b in range(B):
        inds = rois[:, 0].long() == b
        if inds.any():
            r = rois[inds][:, 1:].float()
            pts = rel[inds].clone()
            pts[:, :, 0] = pts[:, :, 0] * (r[:, None, 2] - r[:, None, 0])
            pts[:, :, 1] = pts[:, :, 1] * (r[:, None, 3] - r[:, None, 1])
            pts[:, :, 0] += r[:, None, 0]
            pts[:, :, 1] += r[:, None, 1]
            scale = torch.tensor([W, H], dtype=torch.float32).view(1, 1, 2)
            pts = pts / scale * spatial_scale
            pts = pts.unsqueeze(0)                  # [1, k, S*S, 2]
            o = F.grid_sample(feat[b:b + 1], (pts * 2.0 - 1.0).to(feat.dtype), mode='bilinear',
                              padding_mode='zeros', align_corners=not aligned)
            outs.append(o.squeeze(0).transpose(0, 1))   # [k, C, S*S]
    if not outs:
        return feat.new_zeros((0, C, S, S))
    return torch.cat(outs, dim=0).reshape(K, C, S, S)


# --------------------------------------------------------------------------
# Deformable convolution v1
#   spec: mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:84-115 (bilinear),
#         :190-243 (im2col), deform_conv_cuda.cpp:152-260 (GEMM),
#         mmdet/ops/dcn/deform_conv.py:189-275 (module; no bias)
#   call site: mask_heads/dynamask_head.py:84 (3x3, s1, p1, deform_groups=2)
# --------------------------------------------------------------------------
def deform_im2col(x, offset, kh=3, kw=3, stride=1, pad=1, dil=1, deform_groups=1):
    """x [N,C,H,W], offset [N, dg*2*kh*kw, Ho, Wo] -> columns [N, C, kh*kw, Ho, Wo].

    Differentiable wrt x and offset (bilinear weights are functions of offset).
    """
    N, C, H, W = x.shape
    Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    dg = deform_groups
    cpg = C // dg
    off = offset.view(N, dg, kh * kw, 2, Ho, Wo)
    hs = (torch.arange(Ho, dtype=x.dtype) * stride - pad).view(1, 1, Ho, 1)
    ws = (torch.arange(Wo, dtype=x.dtype) * stride - pad).view(1, 1, 1, Wo)
    xg = x.reshape(N, dg, cpg, H * W)
    cols = []
    for i in range(kh):
        for j in range(kw):
            t = i * kw + j
            h_im = hs + i * dil + off[:, :, t, 0]      # [N,dg,Ho,Wo]
            w_im = ws + j * dil + off[:, :, t, 1]
            valid = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
            h_low = torch.floor(h_im)
            w_low = torch.floor(w_im)
            lh = h_im - h_low
            lw = w_im - w_low
            hh = 1 - lh
            hw = 1 - lw
            h_low = h_low.long()
            w_low = w_low.long()
            h_high = h_low + 1
            w_high = w_low + 1

            def tap(hi, wi, ok):
                ok = ok & valid
                idx = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).view(N, dg, 1, Ho * Wo)
                v = torch.gather(xg, 3, idx.expand(-1, -1, cpg, -1))
                return v * ok.view(N, dg, 1, Ho * Wo).to(x.dtype)

            v1 = tap(h_low, w_low, (h_low >= 0) & (w_low >= 0))
            v2 = tap(h_low, w_high, (h_low >= 0) & (w_high <= W - 1))
            v3 = tap(h_high, w_low, (h_high <= H - 1) & (w_low >= 0))
            v4 = tap(h_high, w_high, (h_high <= H - 1) & (w_high <= W - 1))

            def wv(a):
                return a.view(N, dg, 1, Ho * Wo)
            val = wv(hh * hw) * v1 + wv(hh * lw) * v2 + wv(lh * hw) * v3 + wv(lh * lw) * v4
            cols.append(val.view(N, C, Ho, Wo))
    return torch.stack(cols, dim=2)


def deform_conv2d(x, offset, weight, stride=1, pad=1, dil=1, deform_groups=1, chunk=16):
    """DCNv1 forward (groups=1, no bias)."""
    Cout, Cin, kh, kw = weight.shape
    outs = []
    w2 = weight.reshape(Cout, Cin * kh * kw)
    for s in range(0, x.shape[0], chunk):
        cols = deform_im2col(x[s:s + chunk], offset[s:s + chunk], kh, kw, stride, pad, dil, deform_groups)
        n, _, _, Ho, Wo = cols.shape
        cols = cols.reshape(n, Cin * kh * kw, Ho * Wo)
        outs.append(torch.matmul(w2, cols).view(n, Cout, Ho, Wo))
    if not outs:
        return x.new_zeros((0, Cout, x.shape[2], x.shape[3]))
    return torch.cat(outs, dim=0)


def deform_conv_pack(x, weight, offset_weight, offset_bias, deform_groups=2):
    """DeformConv2dPack.forward: offset = conv_offset(x) (plain 3x3 conv, bias);
    out = deform_conv(x, offset, weight).  (deform_conv.py:263-280)"""
    offset = F.conv2d(x, offset_weight, offset_bias, stride=1, padding=1)
    return deform_conv2d(x, offset, weight, 1, 1, 1, deform_groups)


# --------------------------------------------------------------------------
# CARAFE (mmcv/ops/carafe.py CARAFEPack), call site fcn_mask_head.py:84-87
# --------------------------------------------------------------------------
def carafe_reassemble(x, mask, k, group, scale):
    """out[n,c,y,x] = sum_{i,j<k} x[n,c, y//s + i - k//2, x//s + j - k//2]
                      * mask[n, grp(c)*k*k + i*k + j, y, x]   (zero outside)."""
    N, C, H, W = x.shape
    r = k // 2
    xp = F.pad(x, (r, r, r, r))
    out = x.new_zeros((N, C, H * scale, W * scale))
    cpg = C // group
    m = mask.view(N, group, k * k, H * scale, W * scale)
    for i in range(k):
        for j in range(k):
            patch = xp[:, :, i:i + H, j:j + W]
            patch = patch.repeat_interleave(scale, dim=2).repeat_interleave(scale, dim=3)
            w = m[:, :, i * k + j].repeat_interleave(cpg, dim=1)
            out = out + patch * w
    return out


def carafe_pack(x, comp_w, comp_b, enc_w, enc_b, scale=2, up_kernel=5, up_group=1,
                encoder_kernel=3, encoder_dilation=1):
    comp = F.conv2d(x, comp_w, comp_b)
    pad = int((encoder_kernel - 1) * encoder_dilation / 2)
    mask = F.conv2d(comp, enc_w, enc_b, padding=pad, dilation=encoder_dilation)
    mask = F.pixel_shuffle(mask, scale)
    n, mc, h, w = mask.shape
    mch = int(mc / float(up_kernel * up_kernel))
    mask = F.softmax(mask.view(n, mch, -1, h, w), dim=2).view(n, mc, h, w).contiguous()
    return carafe_reassemble(x, mask, up_kernel, up_group, scale)


# --------------------------------------------------------------------------- COCO RLE
# Third-party arithmetic absent from /root/reference: pycocotools (requirements/runtime.txt,
# call site mmdet/core/mask/utils.py:55-60 `mask_util.encode(np.array(m[:, :, None], order='F'))`).
# Restated from the published cocoapi/common/maskApi.c (rleEncode, rleToString, rleFrString).
# PARITY UNPINNED: the reference holds no golden RLE strings (tests only call encode on zeros).
def rle_counts(mask):
    """rleEncode: run lengths of a [h, w] 0/1 mask walked column-major, first run = zeros."""
    import numpy as np
    t = np.asarray(mask).astype(np.uint8).T.reshape(-1)          # column-major order
    if t.size == 0:
        return [0]
    change = np.flatnonzero(np.concatenate([[t[0] != 0], t[1:] != t[:-1]]))
    # a boundary at 0 (mask starts with 1) yields the leading empty run of zeros
    edges = np.concatenate([[0], change, [t.size]])
    return np.diff(edges).tolist()


def rle_to_string(cnts):
    """rleToString: 5-bit groups, bit 5 = continuation, +48; counts from the 3rd on are
    differences to the count two positions back."""
    out = bytearray()
    for i, c in enumerate(cnts):
        x = int(c)
        if i > 2:
            x -= int(cnts[i - 2])
        more = True
        while more:
            ch = x & 0x1f
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(ch + 48)
    return bytes(out)


def rle_from_string(s):
    """rleFrString (inverse of rle_to_string)."""
    cnts = []
    p = 0
    while p < len(s):
        x = 0
        k = 0
        more = True
        while more:
            c = s[p] - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return cnts


def rle_encode(mask):
    """What pycocotools.mask.encode returns for one [h, w] bitmap."""
    h, w = mask.shape
    return {'size': [int(h), int(w)], 'counts': rle_to_string(rle_counts(mask))}


def rle_decode(rle):
    import numpy as np
    h, w = rle['size']
    cnts = rle_from_string(rle['counts'])
    t = np.zeros(h * w, dtype=np.uint8)
    p, v = 0, 0
    for c in cnts:
        t[p:p + c] = v
        p += c
        v ^= 1
    return t.reshape(w, h).T
