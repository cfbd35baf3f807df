"""TEST INFRASTRUCTURE ONLY (oracle/): numpy restatement of libjpeg(-turbo)'s default decompression path BEHIND the entropy decoder --
what vitcap_amd/csrc/jpeg.hip computes on the device.  Only tests/ may import this module.

The reference decodes JPEGs with cv2.imdecode (src/tools/common.py:23-31, src/data_layer/transform.py:106-136); cv2 and Pillow both wrap
libjpeg-turbo with its defaults (JDCT_ISLOW, do_fancy_upsampling, JCS_RGB output).  libjpeg-turbo is a third-party dependency that is
NOT in /root/reference (Pillow 12.2 bundles its 3.x line here: `PIL.features.version('jpg')`); this file restates its published algorithm:

  jidctint.c  jpeg_idct_islow      dequantise + 8x8 inverse DCT, 13-bit fixed point, two passes, range limit
  jdsample.c  h2v1_fancy_upsample / h2v2_fancy_upsample   triangle-filter chroma upsampling (3/4, 1/4 weights)
  jdmainct.c  context rows         the rows above the first / below the last real row are copies of that row
  jdcolor.c   ycc_rgb_convert      fixed-point YCbCr -> RGB (SCALEBITS = 16)

Pinned by tests/test_jpeg_cpu.py: host entropy decoder (vitcap_amd/libvitcap_jpeg.so) + this back half == Pillow's decoded pixels, bit for
bit, on every sampling mode / size of the test set."""
import numpy as np

CONST_BITS, PASS1_BITS = 13, 2
F_0_298631336, F_0_390180644, F_0_541196100, F_0_765366865 = 2446, 3196, 4433, 6270
F_0_899976223, F_1_175875602, F_1_501321110, F_1_847759065 = 7373, 9633, 12299, 15137
F_1_961570560, F_2_053119869, F_2_562915447, F_3_072711026 = 16069, 16819, 20995, 25172


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n          # arithmetic shift on signed numpy integers


def _idct_1d(i0, i1, i2, i3, i4, i5, i6, i7, shift):
    """One pass of jpeg_idct_islow over 8 inputs (arrays): returns the 8 outputs descaled by `shift`."""
    z2, z3 = i2, i6
    z1 = (z2 + z3) * F_0_541196100
    tmp2 = z1 + z3 * (-F_1_847759065)
    tmp3 = z1 + z2 * F_0_765366865
    tmp0 = (i0 + i4) << CONST_BITS
    tmp1 = (i0 - i4) << CONST_BITS
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = i7, i5, i3, i1
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * F_1_175875602
    tmp0 = tmp0 * F_0_298631336
    tmp1 = tmp1 * F_2_053119869
    tmp2 = tmp2 * F_3_072711026
    tmp3 = tmp3 * F_1_501321110
    z1 = z1 * (-F_0_899976223)
    z2 = z2 * (-F_2_562915447)
    z3 = z3 * (-F_1_961570560) + z5
    z4 = z4 * (-F_0_390180644) + z5
    tmp0 = tmp0 + z1 + z3
    tmp1 = tmp1 + z2 + z4
    tmp2 = tmp2 + z2 + z3
    tmp3 = tmp3 + z1 + z4
    return [_descale(tmp10 + tmp3, shift), _descale(tmp11 + tmp2, shift), _descale(tmp12 + tmp1, shift), _descale(tmp13 + tmp0, shift),
            _descale(tmp13 - tmp0, shift), _descale(tmp12 - tmp1, shift), _descale(tmp11 - tmp2, shift), _descale(tmp10 - tmp3, shift)]


def range_limit(x):
    """sample_range_limit + CENTERJSAMPLE indexed by x & RANGE_MASK (jdmaster.c prepare_range_limit_table)."""
    v = x & 1023
    out = np.where(v < 128, v + 128, np.where(v < 512, 255, np.where(v < 896, 0, v - 896)))
    return out.astype(np.uint8)


def idct_islow(coefs, qt):
    """coefs int16 [nb, 64] natural order, qt uint16 [64] -> uint8 [nb, 8, 8]."""
    c = coefs.astype(np.int64).reshape(-1, 8, 8) * qt.astype(np.int64).reshape(1, 8, 8)
    # pass 1: columns -> workspace
    cols = _idct_1d(*[c[:, r, :] for r in range(8)], CONST_BITS - PASS1_BITS)
    ws = np.stack(cols, axis=1)                        # [nb, 8 rows, 8 cols]
    # pass 2: rows
    rows = _idct_1d(*[ws[:, :, k] for k in range(8)], CONST_BITS + PASS1_BITS + 3)
    return range_limit(np.stack(rows, axis=2))


def plane(info, coefs, c):
    """Component c as a uint8 [blocks_h*8, blocks_w*8] plane (padded size)."""
    bw, bh, b0 = info.blocks_w[c], info.blocks_h[c], info.block0[c]
    qt = np.array(info.qt[c][:], dtype=np.uint16)
    px = idct_islow(coefs[b0 * 64:(b0 + bw * bh) * 64].reshape(-1, 64), qt)
    return px.reshape(bh, bw, 8, 8).transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8)


def h2v1_fancy(inp):
    """inp [rows, w] (w = downsampled_width > 2) -> [rows, 2 w]."""
    x = inp.astype(np.int32)
    w = x.shape[1]
    out = np.empty((x.shape[0], 2 * w), np.int32)
    left = np.concatenate([x[:, :1], x[:, :-1]], 1)
    right = np.concatenate([x[:, 1:], x[:, -1:]], 1)
    out[:, 0::2] = (x * 3 + left + 1) >> 2
    out[:, 1::2] = (x * 3 + right + 2) >> 2
    out[:, 0] = x[:, 0]
    out[:, -1] = x[:, -1]
    return out.astype(np.uint8)


def h2v2_fancy(inp):
    """inp [h, w] (real samples only; w > 2) -> [2 h, 2 w]."""
    x = inp.astype(np.int32)
    h, w = x.shape
    above = np.concatenate([x[:1], x[:-1]], 0)
    below = np.concatenate([x[1:], x[-1:]], 0)
    out = np.empty((2 * h, 2 * w), np.int32)
    for v, other in ((0, above), (1, below)):
        cs = x * 3 + other                                       # "thiscolsum" of every column
        last = np.concatenate([cs[:, :1], cs[:, :-1]], 1)
        nxt = np.concatenate([cs[:, 1:], cs[:, -1:]], 1)
        even = (cs * 3 + last + 8) >> 4
        odd = (cs * 3 + nxt + 7) >> 4
        even[:, 0] = (cs[:, 0] * 4 + 8) >> 4
        odd[:, -1] = (cs[:, -1] * 4 + 7) >> 4
        out[v::2, 0::2] = even
        out[v::2, 1::2] = odd
    return out.astype(np.uint8)


def ycc_to_rgb(y, cb, cr):
    SCALEBITS, HALF = 16, 1 << 15
    fix = lambda v: int(v * (1 << SCALEBITS) + 0.5)
    y, xb, xr = y.astype(np.int64), cb.astype(np.int64) - 128, cr.astype(np.int64) - 128
    r = y + ((fix(1.40200) * xr + HALF) >> SCALEBITS)
    b = y + ((fix(1.77200) * xb + HALF) >> SCALEBITS)
    g = y + (((-fix(0.34414)) * xb + HALF + (-fix(0.71414)) * xr) >> SCALEBITS)
    return np.stack([np.clip(r, 0, 255), np.clip(g, 0, 255), np.clip(b, 0, 255)], axis=-1).astype(np.uint8)


def backhalf(info, coefs):
    """vitcap_jpeg_info + int16 coefficients -> uint8 (H, W, 3) RGB, what Pillow's decoder returns for the same stream."""
    H, W = info.height, info.width
    yp = plane(info, coefs, 0)[:H, :W]
    if info.ncomp == 1:
        return np.repeat(yp[:, :, None], 3, axis=2)
    ch = []
    for c in (1, 2):
        p = plane(info, coefs, c)[:info.samp_h[c], :info.samp_w[c]]
        if info.hs[0] == 2 and info.vs[0] == 2:
            p = h2v2_fancy(p)
        elif info.hs[0] == 2:
            p = h2v1_fancy(p)
        ch.append(p[:H, :W])
    return ycc_to_rgb(yp, ch[0], ch[1])
