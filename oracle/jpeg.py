"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): a CPU restatement of baseline-JPEG decoding as the reference gets it
from Pillow -- ``meerqat/data/loading.py:108-124`` (``load_image``: ``Image.open(path).convert('RGB')``), called per image by
``meerqat/image/embedding.py:127``.

The arithmetic lives in an un-vendored dependency: Pillow 12.2.0 bundles libjpeg-turbo (``PIL.features``: libjpeg 6.2 API,
libjpeg_turbo True) and decodes with the library's defaults -- ``dct_method = JDCT_ISLOW``, ``do_fancy_upsampling = TRUE``,
no scaling, no colour quantisation.  Restated here from the library's published algorithm (file names are libjpeg's):

  * entropy decoding (jdhuff.c): Huffman-coded sequential scans, interleaved or one component per scan, restart intervals,
    0xFF00 byte stuffing; DC prediction per component; ``HUFF_EXTEND``;
  * dequantisation + inverse DCT (jidctint.c ``jpeg_idct_islow``): the 13-bit fixed-point Loeffler-Ligtenberg-Moschytz
    network, columns first (results kept with PASS1_BITS = 2 extra bits), then rows, each ``DESCALE``d with round-half-up, then
    ``range_limit`` (+ 128, clamped to 0 .. 255);
  * chroma upsampling (jdsample.c): ``fullsize``, ``h2v1_fancy`` ((3 a + b + 1) >> 2, (3 a + c + 2) >> 2), ``h2v2_fancy``
    (vertical 3 : 1 sums of the nearer and the farther row, then (3 s + s' + 8) >> 4, (3 s + s'' + 7) >> 4; first / last column
    (4 s + 8) >> 4, (4 s + 7) >> 4), ``h1v2_fancy`` ((3 a + b + 1) >> 2 upwards, + 2 downwards), and the replicating forms taken
    when a component is no wider than two samples; the row above the first and below the last REAL row of a component
    (``downsampled_height``) is that row again (jdmainct.c ``set_wraparound_pointers`` / ``set_bottom_pointers``), the horizontal edge
    cases sit at ``downsampled_width``, not at the block-padded width;
  * colour conversion (jdcolor.c ``build_ycc_rgb_table`` / ``ycc_rgb_convert``): 16-bit fixed-point tables,
    R = y + Cr_r[cr], G = y + ((Cb_g[cb] + Cr_g[cr]) >> 16), B = y + Cb_b[cb], clamped; grey files replicate Y (Pillow's
    ``L`` -> ``RGB``).

Pinned: ``tests/test_jpeg_oracle_cpu.py`` and ``tools/jpeg_pillow_parity.py`` compare ``decode`` bit for bit with Pillow itself
on files written by Pillow's encoder (qualities 5 .. 100, 4:4:4 / 4:2:2 / 4:2:0 / 4:4:0 / grey, optimised tables or not, restart
intervals, sizes that are not multiples of the MCU, down to 1 x 1) -- see ``profiles/r06_notes.md`` section 6 for the count.
Not restated (``Unsupported`` is raised; a product would hand such a file to Pillow): progressive and arithmetic-coded files,
12-bit samples, CMYK / YCCK, RGB-coded files, sampling factors other than the ones above.  (The library's host half also reads
PROGRESSIVE scans; those are pinned without a Python restatement of jdphuff.c: ``decode_staging`` -- this file's inverse DCT,
upsampling and colour conversion on the library's final coefficients -- against Pillow's pixels, tests/test_jpeg_oracle_cpu.py.)

Scalar Python only in the bit reader (a few microseconds per symbol); everything after the coefficients is numpy."""
import numpy as np


class Unsupported(ValueError):
    """The file is a JPEG this restatement does not cover (progressive, CMYK, ...)."""


# zigzag position k -> natural (row-major) position (jutils.c jpeg_natural_order)
NATURAL_ORDER = np.array([
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63],
    dtype=np.int64)

CONST_BITS, PASS1_BITS = 13, 2
FIX_0_298631336, FIX_0_390180644, FIX_0_541196100, FIX_0_765366865 = 2446, 3196, 4433, 6270
FIX_0_899976223, FIX_1_175875602, FIX_1_501321110, FIX_1_847759065 = 7373, 9633, 12299, 15137
FIX_1_961570560, FIX_2_053119869, FIX_2_562915447, FIX_3_072711026 = 16069, 16819, 20995, 25172


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _lll_1d(i0, i1, i2, i3, i4, i5, i6, i7, shift):
    """One 8-point pass of jpeg_idct_islow on int64 arrays; -> the eight outputs, DESCALEd by ``shift``."""
    z1 = (i2 + i6) * FIX_0_541196100
    tmp2 = z1 + i6 * (-FIX_1_847759065)
    tmp3 = z1 + i2 * FIX_0_765366865
    tmp0 = (i0 + i4) << CONST_BITS
    tmp1 = (i0 - i4) << CONST_BITS
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = i7, i5, i3, i1
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * FIX_1_175875602
    t0 = t0 * FIX_0_298631336
    t1 = t1 * FIX_2_053119869
    t2 = t2 * FIX_3_072711026
    t3 = t3 * FIX_1_501321110
    z1 = z1 * (-FIX_0_899976223)
    z2 = z2 * (-FIX_2_562915447)
    z3 = z3 * (-FIX_1_961570560) + z5
    z4 = z4 * (-FIX_0_390180644) + z5
    t0 = t0 + z1 + z3
    t1 = t1 + z2 + z4
    t2 = t2 + z2 + z3
    t3 = t3 + z1 + z4
    return [_descale(v, shift) for v in (tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3)]


def idct_islow(coef):
    """coef int [n, 8, 8] DEQUANTISED coefficients in natural order -> uint8 [n, 8, 8] samples (jidctint.c jpeg_idct_islow;
    its all-AC-zero shortcuts give the values of the full network: DESCALE(dc << 13, 11) = dc << 2 and
    DESCALE(w << 13, 18) = DESCALE(w, 5))."""
    c = coef.astype(np.int64)
    ws = np.stack(_lll_1d(*[c[:, k, :] for k in range(8)], CONST_BITS - PASS1_BITS), axis=1)      # columns: [n, 8 (row), 8 (col)]
    out = np.stack(_lll_1d(*[ws[:, :, k] for k in range(8)], CONST_BITS + PASS1_BITS + 3), axis=2)  # rows
    return np.clip(out + 128, 0, 255).astype(np.uint8)


# ---------------------------------------------------------------------------------------------------------------------------
# upsampling (jdsample.c); planes are uint8 [downsampled_height, downsampled_width]
def _rows_with_context(p):
    """-> (above, below): for every row the row above / below it, the first / last row standing in for what is not there."""
    above = np.concatenate([p[:1], p[:-1]], axis=0)
    below = np.concatenate([p[1:], p[-1:]], axis=0)
    return above, below


def _h2_fancy_from_sums(s, near_mul, r_even, r_odd, shift):
    """The horizontal half of the fancy filters on int32 rows ``s`` [h, w], w > 2 (the caller checked): pixel 2i =
    (near_mul s[i] + s[i-1] + r_even) >> shift, pixel 2i+1 = (near_mul s[i] + s[i+1] + r_odd) >> shift; at the two ends the
    missing neighbour is the sample itself."""
    left = np.concatenate([s[:, :1], s[:, :-1]], axis=1)
    right = np.concatenate([s[:, 1:], s[:, -1:]], axis=1)
    out = np.empty((s.shape[0], 2 * s.shape[1]), dtype=np.int32)
    out[:, 0::2] = (near_mul * s + left + r_even) >> shift
    out[:, 1::2] = (near_mul * s + right + r_odd) >> shift
    return out


def upsample(p, h_expand, v_expand):
    """One component plane to full resolution (before cropping to the image size), as jinit_upsampler picks the method."""
    p32 = p.astype(np.int32)
    h, w = p.shape
    if h_expand == 1 and v_expand == 1:
        return p
    if h_expand == 2 and v_expand == 1:
        if w > 2:   # h2v1_fancy_upsample: first column = the sample itself and (3 a + c + 2) >> 2; written here with the general form
            out = _h2_fancy_from_sums(p32, 3, 1, 2, 2)
            out[:, 0] = p32[:, 0]
            out[:, -1] = p32[:, -1]
            return out.astype(np.uint8)
        return np.repeat(p, 2, axis=1)
    if h_expand == 1 and v_expand == 2:   # h1v2_fancy_upsample
        above, below = _rows_with_context(p32)
        out = np.empty((2 * h, w), dtype=np.int32)
        out[0::2] = (3 * p32 + above + 1) >> 2
        out[1::2] = (3 * p32 + below + 2) >> 2
        return out.astype(np.uint8)
    if h_expand == 2 and v_expand == 2:
        if w > 2:   # h2v2_fancy_upsample
            above, below = _rows_with_context(p32)
            out = np.empty((2 * h, 2 * w), dtype=np.int32)
            out[0::2] = _h2_fancy_from_sums(3 * p32 + above, 3, 8, 7, 4)
            out[1::2] = _h2_fancy_from_sums(3 * p32 + below, 3, 8, 7, 4)
            return out.astype(np.uint8)
        return np.repeat(np.repeat(p, 2, axis=0), 2, axis=1)
    raise Unsupported(f"sampling expansion {h_expand} x {v_expand}")


# ---------------------------------------------------------------------------------------------------------------------------
# colour conversion (jdcolor.c)
def _fix(x):
    return int(x * 65536 + 0.5)


_X = np.arange(256, dtype=np.int64) - 128
CR_R = ((_fix(1.40200) * _X + 32768) >> 16).astype(np.int32)
CB_B = ((_fix(1.77200) * _X + 32768) >> 16).astype(np.int32)
CR_G = (-_fix(0.71414) * _X).astype(np.int32)
CB_G = (-_fix(0.34414) * _X + 32768).astype(np.int32)


def ycc_to_rgb(y, cb, cr):
    y = y.astype(np.int32)
    r = y + CR_R[cr]
    g = y + ((CB_G[cb] + CR_G[cr]) >> 16)
    b = y + CB_B[cb]
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


# ---------------------------------------------------------------------------------------------------------------------------
# the file: markers, tables, scans
class _Huff:
    """One Huffman table as a 16-bit look-ahead: peek 16 bits -> (code length, symbol)."""

    def __init__(self, counts, symbols):
        self.length = np.zeros(1 << 16, dtype=np.uint8)
        self.symbol = np.zeros(1 << 16, dtype=np.uint8)
        code, k = 0, 0
        for ln in range(1, 17):
            for _ in range(counts[ln - 1]):
                lo = code << (16 - ln)
                hi = (code + 1) << (16 - ln)
                if hi > (1 << 16):
                    raise ValueError("bad Huffman table")
                self.length[lo:hi] = ln
                self.symbol[lo:hi] = symbols[k]
                code += 1
                k += 1
            code <<= 1
        self.length = self.length.tolist()
        self.symbol = self.symbol.tolist()


class _Bits:
    """Bit reader over ONE restart segment whose stuffing has been removed; past the end it reads zeros (jdhuff.c warns and
    does the same)."""

    def __init__(self, data):
        self.data = data
        self.pos = 0
        self.buf = 0
        self.cnt = 0

    def _fill(self):
        d, p = self.data, self.pos
        while self.cnt <= 48:
            b = d[p] if p < len(d) else 0
            p += 1
            self.buf = (self.buf << 8) | b
            self.cnt += 8
        self.pos = p

    def decode(self, table):
        if self.cnt < 16:
            self._fill()
        peek = (self.buf >> (self.cnt - 16)) & 0xFFFF
        ln = table.length[peek]
        if ln == 0:
            raise ValueError("bad Huffman code")
        self.cnt -= ln
        self.buf &= (1 << self.cnt) - 1
        return table.symbol[peek]

    def receive_extend(self, s):
        if s == 0:
            return 0
        if self.cnt < s:
            self._fill()
        v = (self.buf >> (self.cnt - s)) & ((1 << s) - 1)
        self.cnt -= s
        self.buf &= (1 << self.cnt) - 1
        return v if v >= (1 << (s - 1)) else v - (1 << s) + 1


def _segments(data, pos):
    """Entropy-coded bytes from ``pos`` to the next marker that is not RSTn: -> (list of unstuffed restart segments, position of
    that marker)."""
    segs, cur = [], bytearray()
    n = len(data)
    while pos < n:
        b = data[pos]
        if b != 0xFF:
            # run of ordinary bytes
            nxt = data.find(b"\xff", pos)
            if nxt < 0:
                nxt = n
            cur += data[pos:nxt]
            pos = nxt
            continue
        if pos + 1 >= n:
            break
        m = data[pos + 1]
        if m == 0x00:
            cur.append(0xFF)
            pos += 2
        elif 0xD0 <= m <= 0xD7:
            segs.append(bytes(cur))
            cur = bytearray()
            pos += 2
        elif m == 0xFF:
            pos += 1   # fill byte
        else:
            break
    segs.append(bytes(cur))
    return segs, pos


def read_coefficients(data):
    """File bytes -> dict(width, height, components = [dict(id, h, v, tq, blocks_w, blocks_h, coef int16 [blocks_h, blocks_w, 64]
    QUANTISED, zigzag order -> stored in NATURAL order)], qt = {tq: int [64] natural order}, adobe_transform, jfif)."""
    if data[:2] != b"\xff\xd8":
        raise ValueError("not a JPEG file")
    pos, n = 2, len(data)
    qt, huff = {}, {}
    frame = None
    restart = 0
    adobe, jfif = None, False
    while pos < n:
        if data[pos] != 0xFF:
            pos += 1
            continue
        m = data[pos + 1]
        pos += 2
        if m == 0xFF:
            pos -= 1
            continue
        if m == 0xD9:
            break
        if m == 0x01 or 0xD0 <= m <= 0xD7:
            continue
        ln = (data[pos] << 8) | data[pos + 1]
        seg = data[pos + 2:pos + ln]
        pos += ln
        if m == 0xDB:
            q = 0
            while q < len(seg):
                pq, tq = seg[q] >> 4, seg[q] & 15
                q += 1
                if pq:
                    vals = [(seg[q + 2 * i] << 8) | seg[q + 2 * i + 1] for i in range(64)]
                    q += 128
                else:
                    vals = list(seg[q:q + 64])
                    q += 64
                t = np.zeros(64, dtype=np.int64)
                t[NATURAL_ORDER] = vals
                qt[tq] = t
        elif m == 0xC4:
            q = 0
            while q < len(seg):
                tc, th = seg[q] >> 4, seg[q] & 15
                counts = list(seg[q + 1:q + 17])
                ns = sum(counts)
                huff[(tc, th)] = _Huff(counts, list(seg[q + 17:q + 17 + ns]))
                q += 17 + ns
        elif m in (0xC0, 0xC1):
            if seg[0] != 8:
                raise Unsupported(f"{seg[0]}-bit samples")
            height, width = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4]
            comps = []
            for i in range(seg[5]):
                cid, hv, tq = seg[6 + 3 * i: 9 + 3 * i]
                comps.append(dict(id=cid, h=hv >> 4, v=hv & 15, tq=tq))
            if height == 0:
                raise Unsupported("DNL-defined height")
            hmax, vmax = max(c["h"] for c in comps), max(c["v"] for c in comps)
            mcux, mcuy = -(-width // (8 * hmax)), -(-height // (8 * vmax))
            for c in comps:
                c["blocks_w"], c["blocks_h"] = mcux * c["h"], mcuy * c["v"]   # padded to whole MCUs (interleaved scans fill all)
                c["coef"] = np.zeros((c["blocks_h"], c["blocks_w"], 64), dtype=np.int16)
                c["pred"] = 0
            frame = dict(width=width, height=height, components=comps, hmax=hmax, vmax=vmax, mcux=mcux, mcuy=mcuy)
        elif m in (0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise Unsupported({0xC2: "progressive"}.get(m, f"SOF marker {m:#x}"))
        elif m == 0xDD:
            restart = (seg[0] << 8) | seg[1]
        elif m == 0xEE and seg[:5] == b"Adobe" and len(seg) >= 12:
            adobe = seg[11]
        elif m == 0xE0 and seg[:5] == b"JFIF\0":
            jfif = True
        elif m == 0xDA:
            if frame is None:
                raise ValueError("SOS before SOF")
            ns = seg[0]
            scan = []
            for i in range(ns):
                cid, tt = seg[1 + 2 * i], seg[2 + 2 * i]
                comp = next(c for c in frame["components"] if c["id"] == cid)
                scan.append((comp, huff[(0, tt >> 4)], huff[(1, tt & 15)]))
            ss, se, ahal = seg[1 + 2 * ns: 4 + 2 * ns]
            if (ss, se, ahal) != (0, 63, 0):
                raise Unsupported("spectral selection / successive approximation")
            segs, pos = _segments(data, pos)
            _decode_scan(frame, scan, restart, segs)
    if frame is None:
        raise ValueError("no frame")
    frame.update(qt=qt, adobe_transform=adobe, jfif=jfif)
    return frame


def _decode_scan(frame, scan, restart, segs):
    width, height, hmax, vmax = frame["width"], frame["height"], frame["hmax"], frame["vmax"]
    if len(scan) > 1:
        units = [(c, dc, ac, [(by, bx) for by in range(c["v"]) for bx in range(c["h"])]) for c, dc, ac in scan]
        nx, ny = frame["mcux"], frame["mcuy"]
        stepx = {id(c): c["h"] for c, _, _ in scan}
        stepy = {id(c): c["v"] for c, _, _ in scan}
    else:   # one component: an MCU is one block, and only the blocks that hold image samples are coded
        c = scan[0][0]
        units = [(c, scan[0][1], scan[0][2], [(0, 0)])]
        nx = -(-(-(-width * c["h"] // hmax)) // 8)
        ny = -(-(-(-height * c["v"] // vmax)) // 8)
        stepx, stepy = {id(c): 1}, {id(c): 1}
    nat = NATURAL_ORDER.tolist()
    seg_i, bits, todo = 0, _Bits(segs[0]), restart
    for my in range(ny):
        for mx in range(nx):
            if restart and todo == 0:
                seg_i += 1
                bits = _Bits(segs[seg_i] if seg_i < len(segs) else b"")
                for c, _, _ in scan:
                    c["pred"] = 0
                todo = restart
            for c, dc, ac, blocks in units:
                for by, bx in blocks:
                    blk = [0] * 64
                    s = bits.decode(dc)
                    c["pred"] += bits.receive_extend(s)
                    blk[0] = c["pred"]
                    k = 1
                    while k < 64:
                        rs = bits.decode(ac)
                        r, s = rs >> 4, rs & 15
                        if s == 0:
                            if r != 15:
                                break
                            k += 16
                            continue
                        k += r
                        if k > 63:
                            break   # corrupt data; jdhuff.c bounds the index the same way
                        blk[nat[k]] = bits.receive_extend(s)
                        k += 1
                    c["coef"][my * stepy[id(c)] + by, mx * stepx[id(c)] + bx] = blk
            todo -= 1


def decode(data):
    """JPEG file bytes -> uint8 [height, width, 3]: what ``np.asarray(Image.open(f).convert('RGB'))`` holds."""
    f = read_coefficients(data)
    comps, width, height = f["components"], f["width"], f["height"]
    if len(comps) not in (1, 3):
        raise Unsupported(f"{len(comps)} components")
    if len(comps) == 3:
        ids = [c["id"] for c in comps]
        # jdapimin.c default_decompress_parms: JFIF -> YCbCr; Adobe -> its transform flag; neither -> YCbCr unless the ids spell RGB
        if (not f["jfif"] and f["adobe_transform"] == 0) or (not f["jfif"] and f["adobe_transform"] is None and ids == [82, 71, 66]):
            raise Unsupported("RGB-coded file")
    planes = []
    for c in comps:
        q = f["qt"][c["tq"]]
        coef = c["coef"].reshape(-1, 64).astype(np.int64) * q[None, :]
        px = idct_islow(coef.reshape(-1, 8, 8)).reshape(c["blocks_h"], c["blocks_w"], 8, 8)
        plane = px.transpose(0, 2, 1, 3).reshape(c["blocks_h"] * 8, c["blocks_w"] * 8)
        dw = -(-width * c["h"] // f["hmax"])    # downsampled_width / _height (jdmaster.c): the REAL samples of the component
        dh = -(-height * c["v"] // f["vmax"])
        if f["hmax"] % c["h"] or f["vmax"] % c["v"]:
            raise Unsupported("fractional sampling ratio")
        up = upsample(plane[:dh, :dw], f["hmax"] // c["h"], f["vmax"] // c["v"])
        planes.append(up[:height, :width])
    if len(planes) == 1:
        return np.repeat(planes[0][:, :, None], 3, axis=2)
    return ycc_to_rgb(*planes)


def decode_staging(buf):
    """The staging area csrc/jpeg.hip's host half writes for one file (include/meerqat_hip.h, mq_jpeg_read_coefficients: a
    512-byte header, then int16 coefficient blocks in natural order) -> uint8 [height, width, 3]: the arithmetic of ``decode`` on
    the LIBRARY's coefficients (CPU tests of the host half; the device half is compared with Pillow on the GPU)."""
    buf = np.asarray(buf, dtype=np.uint8)
    hw = buf[:128].view(np.int32)
    height, width, ncomp, hmax, vmax = (int(v) for v in hw[1:6])
    if int(hw[0]) == 0x20424752:   # MQ_JPEG_MAGIC_RGB: decoded by other means, stored as is
        return buf[512:512 + height * width * 3].reshape(height, width, 3).copy()
    assert int(hw[0]) == 0x4745504A, "not a staging area"
    planes = []
    for c in range(ncomp):
        ch, cv, bw, bh, first, dw, dh = (int(hw[i + c]) for i in (8, 11, 14, 17, 20, 24, 27))
        q = buf[128 + 128 * c:256 + 128 * c].view(np.uint16).astype(np.int64)
        coef = buf[512 + first * 128:512 + (first + bw * bh) * 128].view(np.int16).astype(np.int64) .reshape(-1, 64) * q[None, :]
        px = idct_islow(coef.reshape(-1, 8, 8)).reshape(bh, bw, 8, 8)
        plane = px.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8)
        planes.append(upsample(plane[:dh, :dw], hmax // ch, vmax // cv)[:height, :width])
    if ncomp == 1:
        return np.repeat(planes[0][:, :, None], 3, axis=2)
    return ycc_to_rgb(*planes)
