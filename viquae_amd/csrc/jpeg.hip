// jpeg.hip -- baseline-JPEG decoding split between the host and the GPU: what the reference gets, per image, from
// `Image.open(path).convert('RGB')` (meerqat/data/loading.py:108-124, called by meerqat/image/embedding.py:127).
// Pillow hands the file to libjpeg-turbo; bit for bit the same RGB bytes are produced here by
//
//   host   mq_jpeg_probe / mq_jpeg_read_coefficients: marker parsing + Huffman decoding -- the ONE interleaved scan of a sequential
//          file (jdhuff.c's algorithm: look-ahead tables, HUFF_EXTEND, DC prediction, restart intervals) or all the scans of a
//          progressive one (jdphuff.c: DC / AC first and refinement scans, end-of-band runs, successive approximation) -- the
//          part of a JPEG decoder that is a serial bit stream -- into quantised coefficient blocks inside the batch's staging
//          buffer, behind a 512-byte header (sizes, sampling, quantisation tables);
//   GPU    jpeg_idct_kernel: dequantisation + jidctint.c's ISLOW inverse DCT (13-bit fixed point, columns then rows,
//          round-half-up descales, + 128, clamp), one thread per 8 x 8 block, IN PLACE (a block's 128 coefficient bytes become
//          its 64 sample bytes);
//          jpeg_rgb_kernel: jdsample.c's fancy chroma upsampling (h2v1, h2v2, h1v2 triangle filters with libjpeg's rounding
//          constants and edge rules, replication when a component is no wider than two samples) + jdcolor.c's fixed-point
//          YCbCr -> RGB, one thread per 16 x 2 pixels (two luma blocks' row pair, one chroma block row of a 4:2:0 file), HWC uint8 at the image's offset of the packed source buffer that
//          mq_image_preprocess_u8 reads.
//
// Byte / integer work; the kernels are bound by HBM (a 4:2:0 image: 3 bytes of coefficients read + 1.5 written + 1.5 read + 3
// written per pixel) and are ~2 % of the CLIP tower's time per batch; the host side is what bounds the job (profiles/r06_notes.md
// section 6).  Anything this decoder does not cover -- arithmetic / lossless / 12-bit files, CMYK, sequential files in several scans,
// incomplete or irregular progressions, other sampling factors, and ANY irregularity of the entropy-coded data (a marker inside the scan, a missing EOI, an invalid
// code) -- is declined (MQ_EUNSUPPORTED / MQ_EINVAL) and the caller decodes that file with Pillow as before, so errors and
// warnings stay the reference's.  Oracle: oracle/jpeg.py, pinned against Pillow (tests/test_jpeg_oracle_cpu.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/meerqat_hip.h"

extern "C" void mq_internal_set_hip_error(int e);

namespace {

#define JPG_HIP(call)                                  \
    do {                                               \
        hipError_t _e = (call);                        \
        if (_e != hipSuccess) { mq_internal_set_hip_error((int)_e); return MQ_EHIP; } \
    } while (0)

// header words (int32) in front of an image's coefficient blocks; the quantisation tables follow at byte 128
enum { H_MAGIC = 0, H_HEIGHT, H_WIDTH, H_NCOMP, H_HMAX, H_VMAX, H_MCUX, H_MCUY, H_CH = 8, H_CV = 11, H_BW = 14, H_BH = 17,
       H_FIRST = 20, H_BLOCKS = 23, H_DW = 24, H_DH = 27 };
constexpr int HDR = MQ_JPEG_HEADER_BYTES;

static const uint8_t NATURAL[64 + 16] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                         41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                         30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                         63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};  // (jdhuff.c: a run past 63 stays inside)

// ------------------------------------------------------------------------------------------------------------------
// host: markers and the Huffman scan
constexpr int LOOK = 10;   // bits of look-ahead of the decoding tables
struct Huff {
    // leading LOOK bits -> bits 0-4: code length (0 = the code is longer than LOOK bits), 5-8: run, 9-12: size (the symbol's two
    // nibbles), bit 13: the value's `size` bits were inside the look-ahead too and bits 16-31 hold it, HUFF_EXTENDed
    uint32_t tab[1 << LOOK];
    int32_t maxcode[18], valoff[17];
    uint8_t vals[256];
    bool defined;
};

static bool build_huff(Huff& t, const uint8_t* counts, const uint8_t* symbols, int nsym) {
    memset(&t, 0, sizeof(t));
    uint8_t size[257];
    uint32_t code[257];
    int p = 0;
    for (int l = 1; l <= 16; ++l)
        for (int i = 0; i < counts[l - 1]; ++i) {
            if (p >= 256) return false;
            size[p++] = (uint8_t)l;
        }
    if (p != nsym) return false;
    uint32_t c = 0;
    int si = p ? size[0] : 0;
    for (int k = 0; k < p;) {
        while (k < p && size[k] == si) code[k++] = c++;
        if (c > (1u << si)) return false;   // the codes of one length must fit that length
        c <<= 1;
        ++si;
    }
    int k = 0;
    for (int l = 1; l <= 16; ++l) {
        if (counts[l - 1]) {
            t.valoff[l] = k - (int)code[k];
            k += counts[l - 1];
            t.maxcode[l] = (int)code[k - 1];
        } else t.maxcode[l] = -1;
    }
    t.maxcode[17] = 0x7FFFFFFF;
    memcpy(t.vals, symbols, (size_t)nsym);
    for (int i = 0; i < p; ++i) {
        const int len = size[i];
        if (len > LOOK) continue;
        const int rs = symbols[i], mag = rs & 15;
        const int lo = (int)code[i] << (LOOK - len);
        for (int j = 0; j < (1 << (LOOK - len)); ++j) {
            uint32_t e = (uint32_t)len | ((uint32_t)(rs >> 4) << 5) | ((uint32_t)mag << 9);
            if (mag && len + mag <= LOOK) {
                int v = (j >> (LOOK - len - mag)) & ((1 << mag) - 1);   // the `mag` bits behind the code
                if (v < (1 << (mag - 1))) v += (int)((~0u) << mag) + 1;  // HUFF_EXTEND
                e |= 0x2000u | ((uint32_t)(uint16_t)(int16_t)v << 16);
            }
            t.tab[lo + j] = e;
        }
    }
    t.defined = true;
    return true;
}

struct Bits {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc;   // the low `n` bits are valid
    int n;
    int fake;       // zero bits appended after the real data ran out (a marker or the end of the file)
    inline void fill() {
        if (n > 32) return;   // a symbol and its value take 31 bits at most
        if (p + 4 <= end) {
            const uint32_t w = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
            const uint32_t x = ~w;
            if (((x - 0x01010101u) & ~x & 0x80808080u) == 0) {   // no 0xFF among the four bytes: nothing to unstuff, no marker
                acc = (acc << 32) | w;
                n += 32;
                p += 4;
                return;
            }
        }
        while (n <= 56) {
            unsigned b = 0;
            if (p < end && p[0] != 0xFF) b = *p++;
            else if (p + 1 < end && p[0] == 0xFF && p[1] == 0x00) { b = 0xFF; p += 2; }
            else fake += 8;
            acc = (acc << 8) | b;
            n += 8;
        }
    }
    inline unsigned peek(int k) const { return (unsigned)(acc >> (n - k)) & ((1u << k) - 1); }
    // at the end of a scan or a restart interval: is marker `m` next?  (any number of 0xFF fill bytes may precede a marker)
    inline bool at_marker(int m) {
        while (p + 2 < end && p[0] == 0xFF && p[1] == 0xFF) ++p;
        return p + 1 < end && p[0] == 0xFF && p[1] == m;
    }
    inline void drop(int k) { n -= k; }
};

struct Frame {
    int height, width, ncomp, hmax, vmax, mcux, mcuy;
    int ch[3], cv[3], tq[3], bw[3], bh[3], first[3], id[3], blocks;
};

struct Parsed {
    Frame f;
    uint16_t qt[4][64];   // natural order
    bool have_qt[4];
    Huff dc[4], ac[4];
    int restart;
    bool progressive, jfif, have_frame, geometry_done;
    int adobe;
    size_t pos;           // where the next marker is expected
    // the scan whose header was read last
    int ns, sc[3], td[3], ta[3], ss, se, ah, al;
    const uint8_t* scan;  // its first entropy-coded byte
};

static inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

static int begin(const uint8_t* d, size_t nbytes, Parsed& out) {
    if (!d || nbytes < 4 || d[0] != 0xFF || d[1] != 0xD8) return MQ_EINVAL;
    memset(&out.f, 0, sizeof(out.f));
    memset(out.have_qt, 0, sizeof(out.have_qt));
    for (int i = 0; i < 4; ++i) out.dc[i].defined = out.ac[i].defined = false;
    out.restart = 0;
    out.progressive = out.jfif = out.have_frame = out.geometry_done = false;
    out.adobe = -1;
    out.pos = 2;
    return MQ_OK;
}

// sampling, colour space and block geometry, once the first scan header arrives (every table-independent reason to decline a file)
static int finish_frame(Parsed& out) {
    Frame& f = out.f;
    // colour space as jdapimin.c default_decompress_parms decides it: JFIF -> YCbCr; else Adobe's transform flag; else YCbCr unless
    // the component ids spell "RGB"
    if (f.ncomp == 3) {
        if (!out.jfif && out.adobe >= 0 && out.adobe != 1) return MQ_EUNSUPPORTED;
        if (!out.jfif && out.adobe < 0 && f.id[0] == 'R' && f.id[1] == 'G' && f.id[2] == 'B') return MQ_EUNSUPPORTED;
    }
    if (f.ncomp == 1) {
        f.ch[0] = f.cv[0] = 1;   // a single-component scan is not interleaved: one block per MCU whatever the factors say
    } else {
        if (f.ch[1] != 1 || f.cv[1] != 1 || f.ch[2] != 1 || f.cv[2] != 1) return MQ_EUNSUPPORTED;
        if (f.ch[0] > 2 || f.cv[0] > 2) return MQ_EUNSUPPORTED;
    }
    f.hmax = f.ch[0];
    f.vmax = f.cv[0];
    if ((int64_t)f.height * f.width > (int64_t)MQ_JPEG_MAX_PIXELS) return MQ_EUNSUPPORTED;
    f.mcux = (f.width + 8 * f.hmax - 1) / (8 * f.hmax);
    f.mcuy = (f.height + 8 * f.vmax - 1) / (8 * f.vmax);
    int first = 0;
    for (int c = 0; c < f.ncomp; ++c) {
        f.bw[c] = f.mcux * f.ch[c];
        f.bh[c] = f.mcuy * f.cv[c];
        f.first[c] = first;
        first += f.bw[c] * f.bh[c];
    }
    f.blocks = first;
    out.geometry_done = true;
    return MQ_OK;
}

// Reads segments from out.pos on.  -> MQ_OK at a start-of-scan (its header in `out`, out.scan = the first entropy-coded byte),
// 1 at the end-of-image marker; MQ_EUNSUPPORTED: a valid-looking file of a kind not handled here; MQ_EINVAL: damaged
static int next_scan(const uint8_t* d, size_t nbytes, Parsed& out) {
    size_t pos = out.pos;
    for (;;) {
        if (pos + 2 > nbytes || d[pos] != 0xFF) return MQ_EINVAL;
        const int m = d[pos + 1];
        if (m == 0xFF) { ++pos; continue; }   // fill byte
        if (m == 0xD9) { out.pos = pos; return 1; }
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7) || m == 0x00) return MQ_EINVAL;
        if (pos + 4 > nbytes) return MQ_EINVAL;
        const size_t ln = (size_t)be16(d + pos + 2);
        if (ln < 2 || pos + 2 + ln > nbytes) return MQ_EINVAL;
        const uint8_t* s = d + pos + 4;
        const size_t sl = ln - 2;
        pos += 2 + ln;
        if (m == 0xDB) {
            size_t q = 0;
            while (q < sl) {
                const int pq = s[q] >> 4, t = s[q] & 15;
                ++q;
                if (t > 3 || pq > 1 || q + (pq ? 128u : 64u) > sl) return MQ_EINVAL;
                if (out.have_qt[t] && out.geometry_done) return MQ_EUNSUPPORTED;   // a table replaced between scans: which blocks it applies to is Pillow's business
                for (int i = 0; i < 64; ++i) out.qt[t][NATURAL[i]] = pq ? (uint16_t)be16(s + q + 2 * i) : s[q + i];
                q += pq ? 128 : 64;
                out.have_qt[t] = true;
            }
        } else if (m == 0xC4) {
            size_t q = 0;
            while (q < sl) {
                if (q + 17 > sl) return MQ_EINVAL;
                const int tc = s[q] >> 4, th = s[q] & 15;
                if (tc > 1 || th > 3) return MQ_EINVAL;
                int ns = 0;
                for (int i = 0; i < 16; ++i) ns += s[q + 1 + i];
                if (ns > 256 || q + 17 + ns > sl) return MQ_EINVAL;
                if (!build_huff(tc ? out.ac[th] : out.dc[th], s + q + 1, s + q + 17, ns)) return MQ_EINVAL;
                q += 17 + (size_t)ns;
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
            if (out.have_frame || sl < 6) return MQ_EINVAL;
            Frame& f = out.f;
            if (s[0] != 8) return MQ_EUNSUPPORTED;
            f.height = be16(s + 1);
            f.width = be16(s + 3);
            f.ncomp = s[5];
            if (f.height < 1 || f.width < 1) return MQ_EUNSUPPORTED;   // (height 0 = defined by a DNL marker)
            if (f.ncomp != 1 && f.ncomp != 3) return MQ_EUNSUPPORTED;
            if (sl < 6 + 3u * f.ncomp) return MQ_EINVAL;
            for (int c = 0; c < f.ncomp; ++c) {
                f.id[c] = s[6 + 3 * c];
                f.ch[c] = s[7 + 3 * c] >> 4;
                f.cv[c] = s[7 + 3 * c] & 15;
                f.tq[c] = s[8 + 3 * c];
                if (f.ch[c] < 1 || f.ch[c] > 4 || f.cv[c] < 1 || f.cv[c] > 4 || f.tq[c] > 3) return MQ_EINVAL;
            }
            if (f.ncomp == 3 && (f.id[0] == f.id[1] || f.id[0] == f.id[2] || f.id[1] == f.id[2])) return MQ_EUNSUPPORTED;
            out.have_frame = true;
            out.progressive = m == 0xC2;
        } else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8)) {
            return MQ_EUNSUPPORTED;   // lossless, arithmetic, differential
        } else if (m == 0xDD) {
            if (sl < 2) return MQ_EINVAL;
            out.restart = be16(s);
        } else if (m == 0xE0) {
            if (sl >= 5 && !memcmp(s, "JFIF\0", 5)) out.jfif = true;
        } else if (m == 0xEE) {
            if (sl >= 12 && !memcmp(s, "Adobe", 5)) out.adobe = s[11];
        } else if (m == 0xDA) {
            if (!out.have_frame) return MQ_EINVAL;
            const Frame& f = out.f;
            if (sl < 1) return MQ_EINVAL;
            out.ns = s[0];
            if (out.ns < 1 || out.ns > f.ncomp || sl < 4u + 2u * out.ns) return MQ_EINVAL;
            for (int i = 0; i < out.ns; ++i) {
                int c = 0;
                while (c < f.ncomp && f.id[c] != s[1 + 2 * i]) ++c;
                if (c == f.ncomp || (i && c <= out.sc[i - 1])) return MQ_EUNSUPPORTED;   // components of a scan come in frame order
                out.sc[i] = c;
                out.td[i] = s[2 + 2 * i] >> 4;
                out.ta[i] = s[2 + 2 * i] & 15;
                if (out.td[i] > 3 || out.ta[i] > 3) return MQ_EINVAL;
            }
            const uint8_t* t = s + 1 + 2 * out.ns;
            out.ss = t[0]; out.se = t[1]; out.ah = t[2] >> 4; out.al = t[2] & 15;
            if (!out.geometry_done) {
                const int rc = finish_frame(out);
                if (rc != MQ_OK) return rc;
            }
            out.scan = d + pos;
            out.pos = pos;
            return MQ_OK;
        }
        // every other segment (APPn, COM, ...) is skipped
    }
}

// the headers of a file up to its first scan, and whether this library decodes that kind of file
static int parse(const uint8_t* d, size_t nbytes, Parsed& out) {
    int rc = begin(d, nbytes, out);
    if (rc != MQ_OK) return rc;
    rc = next_scan(d, nbytes, out);
    if (rc == 1) return MQ_EINVAL;   // no scan at all
    if (rc != MQ_OK) return rc;
    const Frame& f = out.f;
    if (!out.progressive) {   // one interleaved scan of everything
        if (out.ns != f.ncomp || out.ss != 0 || out.se != 63 || out.ah != 0 || out.al != 0) return MQ_EUNSUPPORTED;
        for (int c = 0; c < f.ncomp; ++c) {
            if (!out.dc[out.td[c]].defined || !out.ac[out.ta[c]].defined) return MQ_EINVAL;
            if (!out.have_qt[f.tq[c]]) return MQ_EINVAL;
        }
    }
    return MQ_OK;
}

static inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

static size_t staging_bytes(const Frame& f) {
    const size_t coef = (size_t)f.blocks * 128, rgb = align16((size_t)f.height * f.width * 3);
    return HDR + (coef > rgb ? coef : rgb);
}

// a code longer than the look-ahead: -> its symbol, -1 when no code matches
static inline int decode_slow(Bits& b, const Huff& t) {
    const unsigned v = b.peek(16);
    for (int l = LOOK + 1; l <= 16; ++l) {
        const int code = (int)(v >> (16 - l));
        if (code <= t.maxcode[l]) { b.drop(l); return t.vals[(code + t.valoff[l]) & 255]; }
    }
    return -1;
}

static inline int extend(Bits& b, int s) {
    const int v = (int)b.peek(s);
    b.drop(s);
    return v < (1 << (s - 1)) ? v + (int)((~0u) << s) + 1 : v;
}

// one bit / s bits of the scan
static inline int get_bit(Bits& b) { const int v = (int)b.peek(1); b.drop(1); return v; }

static inline int decode_symbol(Bits& b, const Huff& t) {
    const uint32_t e = t.tab[b.peek(LOOK)];
    if (e & 31) { b.drop((int)(e & 31)); return (int)(((e >> 5) & 15) << 4) | (int)((e >> 9) & 15); }
    return decode_slow(b, t);
}

// the 16-bit domain of libjpeg-turbo's vector inverse DCT (see the note in mq_jpeg_read_coefficients), on a finished block
static bool block_in_domain(const int16_t* blk, const uint16_t* qn) {
    unsigned colsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, big = 0;
    for (int i = 0; i < 64; ++i) {
        const int v = blk[i];
        if (!v) continue;
        const unsigned t = (unsigned)(v < 0 ? -v : v) * qn[i];
        big = t > big ? t : big;
        colsum[i & 7] += t;
        if (colsum[i & 7] > 1000000u) return false;
    }
    return !(big > 5890 || colsum[2] > 5890 || colsum[6] > 5890 || colsum[0] + colsum[4] > 5890 || colsum[3] + colsum[7] > 5890 ||
             colsum[1] + colsum[5] > 5890);
}

// Progressive files (jdphuff.c): every scan adds bits to the coefficient blocks (cleared by the caller) -- DC first / refinement scans
// over one or all components, AC first / refinement scans over one component and a band Ss .. Se, successive approximation by Al.
// The progression must be the regular one (each coefficient's first scan with Ah = 0, every later one with Ah = the Al before it,
// down to Al = 0 for all 64 coefficients of every component: libjpeg only warns about anything else, and an INCOMPLETE progression
// makes it smooth across blocks).  -> false: the file is Pillow's.
static bool read_progressive(const uint8_t* file, size_t nbytes, Parsed* ps, int16_t* coef) {
    const Frame& f = ps->f;
    int bits_known[3][64];   // libjpeg's coef_bits: -1 = not seen yet, else the Al of the coefficient's last scan
    for (int c = 0; c < 3; ++c)
        for (int k = 0; k < 64; ++k) bits_known[c][k] = -1;
    for (int scans = 0;; ++scans) {
        if (scans > 256) return false;
        const int ns = ps->ns, ss = ps->ss, se = ps->se, ah = ps->ah, al = ps->al;
        if (ss > se || se > 63 || al > 13) return false;
        if (ss == 0 ? se != 0 : ns != 1) return false;        // DC scans hold the DC only; AC scans one component
        if (ah != 0 && al != ah - 1) return false;
        for (int i = 0; i < ns; ++i) {
            const int c = ps->sc[i];
            if (ss != 0 && bits_known[c][0] < 0) return false;   // AC before the component's DC
            for (int k = ss; k <= se; ++k) {
                if ((bits_known[c][k] < 0 ? 0 : bits_known[c][k]) != ah) return false;
                bits_known[c][k] = al;
            }
            if (ss == 0 ? (ah == 0 && !ps->dc[ps->td[i]].defined) : !ps->ac[ps->ta[i]].defined) return false;
        }
        Bits b{ps->scan, file + nbytes, 0, 0, 0};
        int pred[3] = {0, 0, 0};
        int todo = ps->restart, rst = 0;
        unsigned eobrun = 0;
        // the scan's MCUs: interleaved (ns > 1) = the frame's MCUs; one component = its blocks that hold image samples
        const int c0 = ps->sc[0];
        const int nx = ns > 1 ? f.mcux : ((f.width * f.ch[c0] + f.hmax - 1) / f.hmax + 7) / 8;
        const int ny = ns > 1 ? f.mcuy : ((f.height * f.cv[c0] + f.vmax - 1) / f.vmax + 7) / 8;
        const int p1 = 1 << al, m1 = (int)((~0u) << al);
        for (int my = 0; my < ny; ++my)
            for (int mx = 0; mx < nx; ++mx) {
                if (ps->restart && todo == 0) {
                    if (b.n < b.fake || b.n - b.fake >= 8) return false;
                    if (!b.at_marker(0xD0 + rst)) return false;
                    b.p += 2;
                    b.acc = 0; b.n = 0; b.fake = 0;
                    rst = (rst + 1) & 7;
                    pred[0] = pred[1] = pred[2] = 0;
                    eobrun = 0;
                    todo = ps->restart;
                }
                for (int i = 0; i < ns; ++i) {
                    const int c = ps->sc[i];
                    const int nby = ns > 1 ? f.cv[c] : 1, nbx = ns > 1 ? f.ch[c] : 1;
                    for (int by = 0; by < nby; ++by)
                        for (int bx = 0; bx < nbx; ++bx) {
                            int16_t* blk = coef + ((size_t)f.first[c] + (size_t)(my * nby + by) * f.bw[c] + (mx * nbx + bx)) * 64;
                            b.fill();
                            if (ss == 0) {
                                if (ah == 0) {   // DC, first scan
                                    const int sz = decode_symbol(b, ps->dc[ps->td[i]]);
                                    if (sz < 0 || sz > 11) return false;
                                    if (sz) pred[c] += extend(b, sz);
                                    if (pred[c] < -32768 || pred[c] > 32767) return false;
                                    const int v = (int)((unsigned)pred[c] << al);
                                    if (v < -32768 || v > 32767) return false;
                                    blk[0] = (int16_t)v;
                                } else if (get_bit(b)) blk[0] = (int16_t)(blk[0] | p1);   // DC refinement
                                continue;
                            }
                            const Huff& ac = ps->ac[ps->ta[i]];
                            if (ah == 0) {   // AC band, first scan
                                if (eobrun > 0) { --eobrun; continue; }
                                for (int k = ss; k <= se; ++k) {
                                    b.fill();
                                    const int rs = decode_symbol(b, ac);
                                    if (rs < 0) return false;
                                    const int r = rs >> 4, sz = rs & 15;
                                    if (sz) {
                                        k += r;
                                        if (k > se) return false;
                                        const int v = (int)((unsigned)extend(b, sz) << al);
                                        if (v < -32768 || v > 32767) return false;
                                        blk[NATURAL[k]] = (int16_t)v;
                                    } else if (r == 15) k += 15;
                                    else {
                                        eobrun = 1u << r;
                                        if (r) eobrun += b.peek(r), b.drop(r);
                                        --eobrun;
                                        break;
                                    }
                                }
                                continue;
                            }
                            // AC band, refinement: one more bit for the coefficients already non-zero, new +-1 coefficients between them
                            int k = ss;
                            if (eobrun == 0) {
                                for (; k <= se; ++k) {
                                    b.fill();
                                    const int rs = decode_symbol(b, ac);
                                    if (rs < 0) return false;
                                    int r = rs >> 4, sz = rs & 15, nv = 0;
                                    if (sz) {
                                        if (sz != 1) return false;
                                        nv = get_bit(b) ? p1 : m1;
                                    } else if (r != 15) {
                                        eobrun = 1u << r;
                                        if (r) eobrun += b.peek(r), b.drop(r);
                                        break;   // the rest of the band is handled below
                                    }
                                    do {
                                        int16_t* cf = blk + NATURAL[k];
                                        if (*cf != 0) {
                                            b.fill();
                                            if (get_bit(b) && (*cf & p1) == 0) *cf = (int16_t)(*cf >= 0 ? *cf + p1 : *cf + m1);
                                        } else if (--r < 0) break;
                                        ++k;
                                    } while (k <= se);
                                    if (nv) {
                                        if (k > se) return false;
                                        blk[NATURAL[k]] = (int16_t)nv;
                                    }
                                }
                            }
                            if (eobrun > 0) {
                                for (; k <= se; ++k) {
                                    int16_t* cf = blk + NATURAL[k];
                                    if (*cf != 0) {
                                        b.fill();
                                        if (get_bit(b) && (*cf & p1) == 0) *cf = (int16_t)(*cf >= 0 ? *cf + p1 : *cf + m1);
                                    }
                                }
                                --eobrun;
                            }
                        }
                }
                --todo;
            }
        if (eobrun != 0) return false;   // a run of empty bands that outlives its scan
        // nothing but the padding of the last byte may be left, and a marker follows
        if (b.n < b.fake || b.n - b.fake >= 8) return false;
        ps->pos = (size_t)(b.p - file);
        const int rc = next_scan(file, nbytes, *ps);
        if (rc == 1) break;   // end of image
        if (rc != MQ_OK) return false;
    }
    for (int c = 0; c < f.ncomp; ++c) {
        if (!ps->have_qt[f.tq[c]]) return false;
        for (int k = 0; k < 64; ++k)
            if (bits_known[c][k] != 0) return false;   // an incomplete progression
        // the blocks outside the component's image area but inside the MCU grid got their DC from the interleaved scans only; all
        // blocks must lie in the vector code's domain
        for (int bi = 0; bi < f.bw[c] * f.bh[c]; ++bi)
            if (!block_in_domain(coef + ((size_t)f.first[c] + bi) * 64, ps->qt[f.tq[c]])) return false;
    }
    return true;
}

}  // namespace

extern "C" {

int mq_jpeg_probe(const uint8_t* file_host, size_t nbytes, int64_t* info_host) {
    if (!info_host) return MQ_EINVAL;
    Parsed* ps = new Parsed;
    const int rc = parse(file_host, nbytes, *ps);
    if (rc == MQ_OK) {
        const Frame& f = ps->f;
        info_host[0] = f.height;
        info_host[1] = f.width;
        info_host[2] = f.ncomp;
        info_host[3] = f.blocks;
        info_host[4] = (int64_t)staging_bytes(f);
        info_host[5] = f.ncomp == 1 ? 0 : f.hmax * 16 + f.vmax;
    }
    delete ps;
    return rc;
}

int mq_jpeg_read_coefficients(const uint8_t* file_host, size_t nbytes, void* staging_host, size_t staging_cap) {
    if (!staging_host || (reinterpret_cast<uintptr_t>(staging_host) & 3)) return MQ_EINVAL;
    Parsed* ps = new Parsed;
    int rc = parse(file_host, nbytes, *ps);
    if (rc != MQ_OK) { delete ps; return rc; }
    const Frame& f = ps->f;
    if (staging_bytes(f) > staging_cap) { delete ps; return MQ_EINVAL; }
    int32_t* hw = static_cast<int32_t*>(staging_host);
    memset(hw, 0, HDR);
    hw[H_HEIGHT] = f.height; hw[H_WIDTH] = f.width; hw[H_NCOMP] = f.ncomp; hw[H_HMAX] = f.hmax; hw[H_VMAX] = f.vmax;
    hw[H_MCUX] = f.mcux; hw[H_MCUY] = f.mcuy; hw[H_BLOCKS] = f.blocks;
    uint16_t* q = reinterpret_cast<uint16_t*>(static_cast<uint8_t*>(staging_host) + 128);
    for (int c = 0; c < f.ncomp; ++c) {
        hw[H_CH + c] = f.ch[c]; hw[H_CV + c] = f.cv[c]; hw[H_BW + c] = f.bw[c]; hw[H_BH + c] = f.bh[c]; hw[H_FIRST + c] = f.first[c];
        hw[H_DW + c] = (f.width * f.ch[c] + f.hmax - 1) / f.hmax;    // downsampled_width / _height (jdmaster.c): the REAL samples
        hw[H_DH + c] = (f.height * f.cv[c] + f.vmax - 1) / f.vmax;
        memcpy(q + 64 * c, ps->qt[f.tq[c]], 128);
    }
    int16_t* coef = reinterpret_cast<int16_t*>(static_cast<uint8_t*>(staging_host) + HDR);
    memset(coef, 0, (size_t)f.blocks * 128);

    if (ps->progressive) {
        const bool ok = read_progressive(file_host, nbytes, ps, coef);
        if (ok)   // the tables may have arrived after the first scan header
            for (int c = 0; c < f.ncomp; ++c) memcpy(q + 64 * c, ps->qt[f.tq[c]], 128);
        delete ps;
        if (!ok) return MQ_EINVAL;
        hw[H_MAGIC] = MQ_JPEG_MAGIC_COEFFICIENTS;
        return MQ_OK;
    }
    Bits b{ps->scan, file_host + nbytes, 0, 0, 0};
    int pred[3] = {0, 0, 0};
    int todo = ps->restart, rst = 0;
    bool bad = false;
    for (int my = 0; my < f.mcuy && !bad; ++my)
        for (int mx = 0; mx < f.mcux && !bad; ++mx) {
            if (ps->restart && todo == 0) {
                // the interval's bits are used up: whatever is left of the last byte is padding; the marker must follow at once
                if (b.n < b.fake || b.n - b.fake >= 8) { bad = true; break; }
                if (!b.at_marker(0xD0 + rst)) { bad = true; break; }
                b.p += 2;
                b.acc = 0; b.n = 0; b.fake = 0;
                rst = (rst + 1) & 7;
                pred[0] = pred[1] = pred[2] = 0;
                todo = ps->restart;
            }
            for (int c = 0; c < f.ncomp; ++c) {
                const Huff& dc = ps->dc[ps->td[c]];
                const Huff& ac = ps->ac[ps->ta[c]];
                for (int by = 0; by < f.cv[c]; ++by)
                    for (int bx = 0; bx < f.ch[c]; ++bx) {
                        int16_t* blk = coef + ((size_t)f.first[c] + (size_t)(my * f.cv[c] + by) * f.bw[c] + (mx * f.ch[c] + bx)) * 64;
                        b.fill();
                        {
                            const uint32_t e = dc.tab[b.peek(LOOK)];
                            if (e & 0x2000u) {
                                b.drop((int)(e & 31) + (int)((e >> 9) & 15));
                                pred[c] += (int)(int16_t)(e >> 16);
                            } else {
                                int s;
                                if (e & 31) { b.drop((int)(e & 31)); s = (int)((e >> 9) & 15) | ((int)((e >> 5) & 15) << 4); }
                                else s = decode_slow(b, dc);
                                if (s < 0 || s > 11) { bad = true; goto done; }
                                if (s) pred[c] += extend(b, s);
                            }
                        }
                        blk[0] = (int16_t)pred[c];
                        // per column of the block, the sum of its DEQUANTISED magnitudes: see the note behind the loop
                        const uint16_t* qn = ps->qt[f.tq[c]];
                        if (pred[c] < -32768 || pred[c] > 32767) { bad = true; goto done; }
                        unsigned colsum[8] = {(unsigned)(pred[c] < 0 ? -pred[c] : pred[c]) * qn[0], 0, 0, 0, 0, 0, 0, 0};
                        unsigned big = colsum[0];
                        for (int k = 1; k < 64;) {
                            b.fill();
                            const uint32_t e = ac.tab[b.peek(LOOK)];
                            int run, v;
                            if (e & 0x2000u) {   // code and value inside the look-ahead
                                b.drop((int)(e & 31) + (int)((e >> 9) & 15));
                                run = (int)((e >> 5) & 15);
                                v = (int)(int16_t)(e >> 16);
                            } else {
                                int s;
                                if (e & 31) {
                                    b.drop((int)(e & 31));
                                    run = (int)((e >> 5) & 15);
                                    s = (int)((e >> 9) & 15);
                                } else {
                                    const int rs = decode_slow(b, ac);
                                    if (rs < 0) { bad = true; goto done; }
                                    run = rs >> 4;
                                    s = rs & 15;
                                }
                                if (s == 0) {
                                    if (run != 15) break;   // end of block
                                    k += 16;
                                    continue;
                                }
                                v = extend(b, s);
                            }
                            k += run;
                            if (k > 63) { bad = true; goto done; }   // (libjpeg warns and carries on: such a file is Pillow's to decode)
                            const int at = NATURAL[k++];
                            blk[at] = (int16_t)v;
                            { const unsigned t = (unsigned)(v < 0 ? -v : v) * qn[at]; big = t > big ? t : big; colsum[at & 7] += t; }
                        }
                        // libjpeg-turbo's vector code keeps the inverse DCT's operands and the column pass's results in 16-bit
                        // lanes (and wraps or saturates there); the C code it replaces, and jpeg_idct_kernel below, work in 32
                        // bits.  They agree while nothing leaves 16 bits.  Column j's results are sums of its eight operands
                        // weighted by 1 (row 0) or sqrt(2) cos(.) <= 1.39, times 2^PASS1_BITS: at most 5.56 colsum[j]; the
                        // row pass adds columns 0 + 4, 3 + 7 and 1 + 5 in 16 bits (the column pass the same rows of operands,
                        // which are smaller).  32767 / 5.56 = 5893.  A block of an encoded picture stays far below (a
                        // full-contrast edge along the block: ~2700 in one column); a file with a block beyond the bound
                        // (damaged data, as a rule) is Pillow's.
                        if (big > 5890 || colsum[2] > 5890 || colsum[6] > 5890 || colsum[0] + colsum[4] > 5890 ||
                            colsum[3] + colsum[7] > 5890 || colsum[1] + colsum[5] > 5890) {
                            bad = true;
                            goto done;
                        }
                    }
            }
            --todo;
        }
done:
    // nothing but the padding of the last byte may be left, and the next thing in the file is the end-of-image marker
    if (!bad && (b.n < b.fake || b.n - b.fake >= 8)) bad = true;
    if (!bad && !b.at_marker(0xD9)) bad = true;
    delete ps;
    if (bad) return MQ_EINVAL;
    hw[H_MAGIC] = MQ_JPEG_MAGIC_COEFFICIENTS;
    return MQ_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// device
namespace {

constexpr int CONST_BITS = 13, PASS1_BITS = 2;

template <int SHIFT>
__device__ __forceinline__ void lll8(int& v0, int& v1, int& v2, int& v3, int& v4, int& v5, int& v6, int& v7) {
    // one 8-point pass of jidctint.c jpeg_idct_islow (even part, odd part, butterflies), DESCALEd by SHIFT
    int z1 = (v2 + v6) * 4433;
    const int tmp2 = z1 + v6 * (-15137);
    const int tmp3 = z1 + v2 * 6270;
    const int tmp0 = (v0 + v4) << CONST_BITS;
    const int tmp1 = (v0 - v4) << CONST_BITS;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    int t0 = v7, t1 = v5, t2 = v3, t3 = v1;
    z1 = t0 + t3;
    int z2 = t1 + t2, z3 = t0 + t2, z4 = t1 + t3;
    const int z5 = (z3 + z4) * 9633;
    t0 *= 2446; t1 *= 16819; t2 *= 25172; t3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5; z4 += z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    constexpr int R = 1 << (SHIFT - 1);
    v0 = (tmp10 + t3 + R) >> SHIFT; v7 = (tmp10 - t3 + R) >> SHIFT;
    v1 = (tmp11 + t2 + R) >> SHIFT; v6 = (tmp11 - t2 + R) >> SHIFT;
    v2 = (tmp12 + t1 + R) >> SHIFT; v5 = (tmp12 - t1 + R) >> SHIFT;
    v3 = (tmp13 + t0 + R) >> SHIFT; v4 = (tmp13 - t0 + R) >> SHIFT;
}

// grid (block chunks, images), 128 threads: thread = one 8 x 8 block of the image, all components in one index space
__global__ __launch_bounds__(128) void jpeg_idct_kernel(uint8_t* __restrict__ buf, const int64_t* __restrict__ items) {
    const int img = blockIdx.y;
    uint8_t* st = buf + items[2 * img];
    const int32_t* hw = reinterpret_cast<const int32_t*>(st);
    if (hw[H_MAGIC] != MQ_JPEG_MAGIC_COEFFICIENTS) return;
    const int nblocks = hw[H_BLOCKS];
    const int blk = blockIdx.x * 128 + threadIdx.x;
    if (blk >= nblocks) return;
    const int ncomp = hw[H_NCOMP];
    int c = 0;
    if (ncomp == 3) c = blk >= hw[H_FIRST + 2] ? 2 : (blk >= hw[H_FIRST + 1] ? 1 : 0);
    const uint16_t* qt = reinterpret_cast<const uint16_t*>(st + 128) + 64 * c;
    int4* p = reinterpret_cast<int4*>(st + HDR + (size_t)blk * 128);
    int v[8][8];   // [row][column], natural order
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int4 cw = p[r];
        const int4 qw = *reinterpret_cast<const int4*>(qt + 8 * r);
        const int cc[4] = {cw.x, cw.y, cw.z, cw.w}, qq[4] = {qw.x, qw.y, qw.z, qw.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[r][2 * j] = (int)(short)(cc[j] & 0xFFFF) * (qq[j] & 0xFFFF);
            v[r][2 * j + 1] = (cc[j] >> 16) * (int)((unsigned)qq[j] >> 16);
        }
    }
#pragma unroll
    for (int col = 0; col < 8; ++col)
        lll8<CONST_BITS - PASS1_BITS>(v[0][col], v[1][col], v[2][col], v[3][col], v[4][col], v[5][col], v[6][col], v[7][col]);
    uint2* o = reinterpret_cast<uint2*>(p);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        lll8<CONST_BITS + PASS1_BITS + 3>(v[r][0], v[r][1], v[r][2], v[r][3], v[r][4], v[r][5], v[r][6], v[r][7]);
        unsigned w[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int s = v[r][j] + 128;
            s = s < 0 ? 0 : (s > 255 ? 255 : s);
            w[j >> 2] |= (unsigned)s << (8 * (j & 3));
        }
        o[r] = make_uint2(w[0], w[1]);   // the block's samples, 8 bytes per row, in the first half of its own coefficient bytes
    }
}

struct Plane {
    const uint8_t* base;  // the component's first block
    int bw, dw, dh;
    __device__ __forceinline__ int at(int y, int x) const {
        return base[((size_t)(y >> 3) * bw + (x >> 3)) * 128 + ((y & 7) << 3) + (x & 7)];
    }
};

__device__ __forceinline__ int upsampled(const Plane& p, int hexp, int vexp, int y, int x) {
    if (hexp == 1 && vexp == 1) return p.at(y, x);
    if (hexp == 2 && vexp == 1) {   // h2v1_fancy_upsample / h2v1_upsample
        const int cx = x >> 1;
        if (p.dw <= 2) return p.at(y, cx);
        const int nb = (x & 1) ? (cx + 1 < p.dw ? cx + 1 : cx) : (cx > 0 ? cx - 1 : 0);
        return (3 * p.at(y, cx) + p.at(y, nb) + ((x & 1) ? 2 : 1)) >> 2;
    }
    const int cy = y >> 1;
    const int ny = (y & 1) ? (cy + 1 < p.dh ? cy + 1 : cy) : (cy > 0 ? cy - 1 : 0);   // the row above the first / below the last real row is that row
    if (hexp == 1) return (3 * p.at(cy, x) + p.at(ny, x) + ((y & 1) ? 2 : 1)) >> 2;    // h1v2_fancy_upsample
    const int cx = x >> 1;
    if (p.dw <= 2) return p.at(cy, cx);                                                 // h2v2_upsample
    const int nx = (x & 1) ? (cx + 1 < p.dw ? cx + 1 : cx) : (cx > 0 ? cx - 1 : 0);
    const int s0 = 3 * p.at(cy, cx) + p.at(ny, cx), s1 = 3 * p.at(cy, nx) + p.at(ny, nx);
    return (3 * s0 + s1 + ((x & 1) ? 7 : 8)) >> 4;                                     // h2v2_fancy_upsample
}

__device__ __forceinline__ unsigned ycc_rgb(int Y, int cb, int cr) {
    // jdcolor.c: FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554, ONE_HALF = 32768
    cb -= 128;
    cr -= 128;
    int r = Y + ((91881 * cr + 32768) >> 16);
    int g = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    int b = Y + ((116130 * cb + 32768) >> 16);
    r = r < 0 ? 0 : (r > 255 ? 255 : r);
    g = g < 0 ? 0 : (g > 255 ? 255 : g);
    b = b < 0 ? 0 : (b > 255 ? 255 : b);
    return (unsigned)r | ((unsigned)g << 8) | ((unsigned)b << 16);
}

struct __attribute__((packed, aligned(4))) Words12 { unsigned w[12]; };

// up to sixteen pixels (24-bit values) of one row -> 3 n consecutive bytes at dst + at
__device__ __forceinline__ void store_row(uint8_t* dst, size_t at, const unsigned (&p)[16], int n) {
    if (n == 16 && !(at & 3)) {
        Words12 o;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            o.w[3 * g] = p[4 * g] | (p[4 * g + 1] << 24);
            o.w[3 * g + 1] = (p[4 * g + 1] >> 8) | (p[4 * g + 2] << 16);
            o.w[3 * g + 2] = (p[4 * g + 2] >> 16) | (p[4 * g + 3] << 8);
        }
        *reinterpret_cast<Words12*>(dst + at) = o;
        return;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < n) { dst[at + 3 * i] = (uint8_t)p[i]; dst[at + 3 * i + 1] = (uint8_t)(p[i] >> 8); dst[at + 3 * i + 2] = (uint8_t)(p[i] >> 16); }
}

// ten chroma samples of one row around the strip's eight (columns cx - 1 .. cx + 8, those beyond the component's REAL samples
// replaced by the edge one): the eight are one block row = ONE 8-byte load; the two neighbours are the last / first byte of the
// block rows the adjacent lanes just loaded (their strips are the adjacent blocks) and come over the lane crossbar -- a byte load
// only where the adjacent lane holds something else (the wave's first / last lane)
__device__ __forceinline__ void chroma_row(const Plane& pl, int row, int sx, bool left_lane, bool right_lane, int (&v)[10]) {
    const int cx = 8 * sx;
    const uint8_t* rp = pl.base + ((size_t)(row >> 3) * pl.bw + sx) * 128 + ((row & 7) << 3);
    const uint2 m = *reinterpret_cast<const uint2*>(rp);
    const unsigned from_left = (unsigned)__shfl_up((int)m.y, 1), from_right = (unsigned)__shfl_down((int)m.x, 1);
    const int e = pl.dw - 1 - cx;   // index of the last real sample among the eight (>= 0)
    int s[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { s[i] = (m.x >> (8 * i)) & 255; s[4 + i] = (m.y >> (8 * i)) & 255; }
    int last = s[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) last = i <= e ? s[i] : last;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[1 + i] = i <= e ? s[i] : last;
    v[0] = sx == 0 ? s[0] : (left_lane ? (int)(from_left >> 24) : (int)rp[-121]);            // byte 7 of the block row to the left
    v[9] = e < 8 ? last : (right_lane ? (int)(from_right & 255) : (int)rp[128]);              // byte 0 of the block row to the right
}

// grid (chunks of 256 strips, images), 256 threads: thread = a 16 x 2 strip of output pixels (x a multiple of 16, y even).  For
// 4:2:0 files -- the usual case -- that is two luma blocks' width and ONE chroma block's: three rows of ten chroma samples per plane
// come in as one 8-byte load each plus two values from the adjacent lanes, the vertical 3 : 1 sums are shared by all the strip's
// pixels, the luma is four 8-byte loads, and a row of the strip leaves as ONE 48-byte store when its address is 4-byte aligned
// (always, when the width is a multiple of 4).  12 memory instructions per 32 pixels.  History, all bit-identical, per 3072-image
// batch: one thread per pixel (nine single-byte loads, three single-byte stores each) 3.79 ms; per 2 x 2 pixels 2.85 ms; per 8 x 2
// pixels (4-byte chroma loads + byte loads for the neighbours) 1.58 ms -- the kernel is bound by the NUMBER of narrow accesses.
// Every other sampling takes the general per-pixel form of the filters.
__global__ __launch_bounds__(256) void jpeg_rgb_kernel(uint8_t* __restrict__ buf, const int64_t* __restrict__ items) {
    const int img = blockIdx.y;
    const uint8_t* st = buf + items[2 * img];
    uint8_t* dst = buf + items[2 * img + 1];
    const int32_t* hw = reinterpret_cast<const int32_t*>(st);
    const int h = hw[H_HEIGHT], w = hw[H_WIDTH];
    const int sw = (w + 15) >> 4, qh = (h + 1) >> 1;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= sw * qh) return;
    const int lane = threadIdx.x & 63;
    const int qy = t / sw, sx = t - qy * sw;
    const int y = 2 * qy, x = 16 * sx;
    const int n = w - x < 16 ? w - x : 16;   // pixels of the strip inside the image
    const bool below = y + 1 < h;
    const size_t at0 = ((size_t)y * w + x) * 3, at1 = at0 + (size_t)w * 3;
    const int magic = hw[H_MAGIC];
    if (magic == MQ_JPEG_MAGIC_RGB) {   // a file Pillow decoded on the host: its RGB bytes sit behind the header
        const uint8_t* s = st + HDR;
        for (int i = 0; i < 3 * n; ++i) {
            dst[at0 + i] = s[at0 + i];
            if (below) dst[at1 + i] = s[at1 + i];
        }
        return;
    }
    if (magic != MQ_JPEG_MAGIC_COEFFICIENTS) return;
    const uint8_t* blocks = st + HDR;
    const int ncomp = hw[H_NCOMP];
    const Plane py{blocks, hw[H_BW], hw[H_DW], hw[H_DH]};
    // luma: two blocks' rows y and y + 1 (y is even: the same blocks; blocks are padded to MCUs: the row exists even when the
    // image ends at y).  The second block may lie beyond the component when the sampling is not 2 x 2: its pixels are not stored.
    const int b0 = 2 * sx, b1 = 2 * sx + 1 < py.bw ? 2 * sx + 1 : b0;
    const uint8_t* yrow = blocks + (size_t)(y >> 3) * py.bw * 128 + ((y & 7) << 3);
    const uint2 ya0 = *reinterpret_cast<const uint2*>(yrow + (size_t)b0 * 128), ya1 = *reinterpret_cast<const uint2*>(yrow + (size_t)b1 * 128);
    const uint2 yb0 = *reinterpret_cast<const uint2*>(yrow + (size_t)b0 * 128 + 8), yb1 = *reinterpret_cast<const uint2*>(yrow + (size_t)b1 * 128 + 8);
    int Y0[16], Y1[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        Y0[i] = (ya0.x >> (8 * i)) & 255; Y0[4 + i] = (ya0.y >> (8 * i)) & 255; Y0[8 + i] = (ya1.x >> (8 * i)) & 255; Y0[12 + i] = (ya1.y >> (8 * i)) & 255;
        Y1[i] = (yb0.x >> (8 * i)) & 255; Y1[4 + i] = (yb0.y >> (8 * i)) & 255; Y1[8 + i] = (yb1.x >> (8 * i)) & 255; Y1[12 + i] = (yb1.y >> (8 * i)) & 255;
    }
    unsigned p0[16], p1[16];
    if (ncomp == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { p0[i] = Y0[i] * 0x010101u; p1[i] = Y1[i] * 0x010101u; }
    } else {
        const int hexp = hw[H_HMAX], vexp = hw[H_VMAX];   // chroma is 1 x 1: its expansion is the luma's factor
        const Plane pb{blocks + (size_t)hw[H_FIRST + 1] * 128, hw[H_BW + 1], hw[H_DW + 1], hw[H_DH + 1]};
        const Plane pr{blocks + (size_t)hw[H_FIRST + 2] * 128, hw[H_BW + 2], hw[H_DW + 2], hw[H_DH + 2]};
        if (hexp == 2 && vexp == 2 && pb.dw > 2) {
            // h2v2_fancy_upsample: chroma row qy between its neighbours (beyond the component's REAL rows: the edge row).  The
            // adjacent lane holds the adjacent strip of the same rows unless this lane is the wave's first / last one or the
            // strip the row's (the lanes of a wave that are past the image's last strip have left: never the right neighbour of a
            // strip with sx + 1 < sw)
            const bool left_lane = lane > 0 && sx > 0, right_lane = lane < 63 && sx + 1 < sw;
            const int r0 = qy > 0 ? qy - 1 : 0, r2 = qy + 1 < pb.dh ? qy + 1 : qy;
            int c0[2][16], c1[2][16];   // [Cb, Cr][pixel] of the upper / the lower row
#pragma unroll
            for (int comp = 0; comp < 2; ++comp) {
                const Plane& pl = comp ? pr : pb;
                int a[10], m[10], d[10];
                chroma_row(pl, r0, sx, left_lane, right_lane, a);
                chroma_row(pl, qy, sx, left_lane, right_lane, m);
                chroma_row(pl, r2, sx, left_lane, right_lane, d);
                int tu[10], tl[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) { tu[i] = 3 * m[i] + a[i]; tl[i] = 3 * m[i] + d[i]; }   // the row above is the farther one for the upper output row
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    c0[comp][2 * j] = (3 * tu[j + 1] + tu[j] + 8) >> 4;
                    c0[comp][2 * j + 1] = (3 * tu[j + 1] + tu[j + 2] + 7) >> 4;
                    c1[comp][2 * j] = (3 * tl[j + 1] + tl[j] + 8) >> 4;
                    c1[comp][2 * j + 1] = (3 * tl[j + 1] + tl[j + 2] + 7) >> 4;
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) { p0[i] = ycc_rgb(Y0[i], c0[0][i], c0[1][i]); p1[i] = ycc_rgb(Y1[i], c1[0][i], c1[1][i]); }
        } else {
            const int y1 = below ? y + 1 : y;
#pragma unroll 1
            for (int i = 0; i < 16; ++i) {   // (clamped: the values of absent pixels are not stored)
                const int xi = x + i < w ? x + i : w - 1;
                const unsigned u = ycc_rgb(py.at(y, xi), upsampled(pb, hexp, vexp, y, xi), upsampled(pr, hexp, vexp, y, xi));
                const unsigned l = ycc_rgb(py.at(y1, xi), upsampled(pb, hexp, vexp, y1, xi), upsampled(pr, hexp, vexp, y1, xi));
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k == i) { p0[k] = u; p1[k] = l; }
            }
        }
    }
    store_row(dst, at0, p0, n);
    if (below) store_row(dst, at1, p1, n);
}

}  // namespace

extern "C" int mq_jpeg_decode_rgb_u8(uint8_t* buf_dev, const int64_t* items_dev, int n_images, int max_blocks, int64_t max_strips,
                                     void* stream) {
    if (n_images == 0) return MQ_OK;
    if (!buf_dev || !items_dev || n_images < 0 || max_blocks < 0 || max_strips < 1 || max_strips > MQ_JPEG_MAX_PIXELS ||
        n_images > 65535 || (reinterpret_cast<uintptr_t>(buf_dev) & 15))
        return MQ_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (max_blocks > 0)
        hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((max_blocks + 127) / 128), (unsigned)n_images), dim3(128), 0, st, buf_dev, items_dev);
    hipLaunchKernelGGL(jpeg_rgb_kernel, dim3((unsigned)((max_strips + 255) / 256), (unsigned)n_images), dim3(256), 0, st, buf_dev, items_dev);
    JPG_HIP(hipGetLastError());
    return MQ_OK;
}
