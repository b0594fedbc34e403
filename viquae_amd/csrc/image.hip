// image.hip -- image preprocessing in front of the CLIP tower on gfx950: what
// `transform(images, return_tensors="pt")` computes in meerqat/image/embedding.py:141-152 with the
// CLIPFeatureExtractor of experiments/image_embedding/clip/vit_config.json:13-17 -- Pillow's 8-bit
// `Image.resize(size, BICUBIC)` (ImagingResample: double-precision filter weights rounded to 22-bit fixed point,
// a horizontal then a vertical pass of integer multiply-adds, each rounded and clipped to uint8), the centre crop,
// `float32(float64(u8) * rescale_factor)` and `(x - mean) / std`.  C ABI: include/meerqat_hip.h.
//
// The decoded images of a batch arrive as ONE packed uint8 buffer (HWC, RGB, each image at its own offset) with a
// small geometry table (mq_image_plan, host arithmetic inside this library).  Three launches per batch:
//
//   resample_coeffs_kernel   per (image, axis, output coordinate inside the crop window): bounds + fixed-point taps,
//                            f64 without contraction = the C doubles of Pillow's precompute_coeffs
//   resample_rows_kernel     horizontal pass, only the source rows the crop window's vertical taps touch and only
//                            the crop window's columns: uint8 [rows][crop_w][3] per image
//   resample_cols_kernel     vertical pass + rescale + normalise (a 3 x 256 table) + HWC -> CHW:
//                            float32 [B][3][crop_h][crop_w]
//
// Byte/integer work bound by HBM (each source pixel is read once from HBM, taps hit L1/L2); bit-exact with Pillow
// + transformers by construction (integer MACs; the float steps are single IEEE operations).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/meerqat_hip.h"

extern "C" void mq_internal_set_hip_error(int e);

namespace {

#define IMG_HIP(call)                                  \
    do {                                               \
        hipError_t _e = (call);                        \
        if (_e != hipSuccess) { mq_internal_set_hip_error((int)_e); return MQ_EHIP; } \
    } while (0)

constexpr int PRECISION_BITS = 32 - 8 - 2;  // Resample.c
constexpr int G = MQ_IMAGE_GEOM;            // int64 fields per image, see mq_image_plan
enum { G_SRC = 0, G_INH, G_INW, G_OUTH, G_OUTW, G_TOP, G_LEFT, G_COEFH, G_COEFV, G_INTER, G_KH, G_KV };

__host__ __device__ inline double filter_support(int filter) { return filter == MQ_IMAGE_BICUBIC ? 2.0 : 1.0; }

// ksize of Pillow's precompute_coeffs for the full-image box
__host__ __device__ inline int resample_ksize(int in_size, int out_size, int filter) {
    double scale = (double)((float)in_size - 0.0f) / out_size;
    double filterscale = scale < 1.0 ? 1.0 : scale;
    return (int)ceil(filter_support(filter) * filterscale) * 2 + 1;
}

__device__ __forceinline__ double filter_eval(int filter, double x) {
    if (x < 0.0) x = -x;
    if (filter == MQ_IMAGE_BICUBIC) {  // Keys, a = -0.5
        if (x < 1.0) return ((-0.5 + 2.0) * x - (-0.5 + 3.0)) * x * x + 1;
        if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * -0.5;
        return 0.0;
    }
    return x < 1.0 ? 1.0 - x : 0.0;
}

struct ImgArgs {
    const uint8_t* src;
    const int64_t* geom;  // [B][G]
    int32_t* bounds;      // [B][2 axes][crop_max][2] (first source index, taps)
    int32_t* coefs;       // per image / axis at geom[G_COEF*], [crop][ksize]
    uint8_t* inter;       // per image at geom[G_INTER]: [rows][crop_w][3]
    int32_t* rowspan;     // [B][2]: first source row, rows the vertical pass touches
    float* out;           // [B][3][crop_h][crop_w]
    float* lut;           // [3][256] u8 -> float (rescale, normalise)
    int B, crop_h, crop_w, crop_max, filter, flags, rpb, row_pitch;
    double rescale;
    float mean[3], stdv[3];
};

// One thread per (image, axis, j): the taps of output coordinate crop_offset + j.
__global__ __launch_bounds__(256) void resample_coeffs_kernel(const ImgArgs a) {
    const int b = blockIdx.y, axis = blockIdx.z;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int64_t* g = a.geom + (size_t)b * G;
    const int crop = axis ? a.crop_h : a.crop_w;
    if (j >= crop) return;
    const int in_size = (int)(axis ? g[G_INH] : g[G_INW]);
    const int out_size = (int)(axis ? g[G_OUTH] : g[G_OUTW]);
    const int xx = (int)(axis ? g[G_TOP] : g[G_LEFT]) + j;
    const int ksize = (int)(axis ? g[G_KV] : g[G_KH]);
    // horizontal taps are stored [tap][column] (the row pass runs one thread per column: coalesced), vertical
    // taps [row][tap] (the column pass reads them wave-uniformly)
    int32_t* k = a.coefs + (axis ? g[G_COEFV] + (size_t)j * ksize : g[G_COEFH] + j);
    const int kstride = axis ? 1 : crop;
    int32_t* bd = a.bounds + (((size_t)b * 2 + axis) * a.crop_max + j) * 2;

    const double scale = (double)((float)in_size - 0.0f) / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = filter_support(a.filter) * filterscale;
    const double center = 0.0 + (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += filter_eval(a.filter, (x + xmin - center + 0.5) * ss);
    for (int x = 0; x < xmax; ++x) {
        double w = filter_eval(a.filter, (x + xmin - center + 0.5) * ss);
        if (ww != 0.0) w /= ww;
        k[(size_t)x * kstride] = w < 0 ? (int)(-0.5 + w * (1 << PRECISION_BITS)) : (int)(0.5 + w * (1 << PRECISION_BITS));
    }
    for (int x = xmax; x < ksize; ++x) k[(size_t)x * kstride] = 0;
    bd[0] = xmin;
    bd[1] = xmax;
    if (axis == 1 && (j == 0 || j == crop - 1)) {
        // source rows the vertical pass touches: [first tap of the crop's first row, last tap of its last row)
        if (j == 0) a.rowspan[b * 2 + 0] = xmin;
        if (j == crop - 1) a.rowspan[b * 2 + 1] = xmin + xmax;  // end (exclusive); turned into a count by the reader
    }
}

__device__ __forceinline__ unsigned lds_byte_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    // Opaque to the optimiser on purpose: shift + clamp pairs are otherwise fused into v_ashr_pk_u8_i32, whose result the
    // compiler then ORs into a dword as if the upper 16 bits of the destination were zero -- on gfx950 they keep the
    // register's previous contents (observed: bytes 2-3 of every packed dword polluted by accumulator bits).
    asm volatile("" : "+v"(v));
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

constexpr int ROWS_PER_BLOCK = 8;

// Horizontal pass.  blockIdx.y = image, blockIdx.x = group of `rpb` (<= 8) needed source rows.  The group's rows --
// only the byte span the crop window's taps touch -- are staged in LDS with aligned 16-byte loads (a row starts at
// any byte: its LDS image keeps the row's offset inside its first 16-byte line); then one thread per crop column keeps
// that column's taps in flight for all rows of the group.
__global__ __launch_bounds__(256) void resample_rows_kernel(const ImgArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t rows_lds[];
    const int b = blockIdx.y;
    const int64_t* g = a.geom + (size_t)b * G;
    const int row0 = a.rowspan[b * 2 + 0], nrows = a.rowspan[b * 2 + 1] - row0;
    const int r0 = blockIdx.x * a.rpb;
    if (r0 >= nrows) return;
    const int in_w = (int)g[G_INW];
    const int32_t* coef = a.coefs + g[G_COEFH];
    const int32_t* bd = a.bounds + ((size_t)b * 2 + 0) * a.crop_max * 2;
    uint8_t* dst = a.inter + g[G_INTER];
    const int rows = min(a.rpb, nrows - r0);
    const int lo = bd[0] * 3, hi = (bd[(a.crop_w - 1) * 2] + bd[(a.crop_w - 1) * 2 + 1]) * 3;  // byte span of a row
    const int pitch = a.row_pitch;  // multiple of 16, >= span + 16
    uint8_t* tile = rows_lds + (size_t)a.rpb * pitch;  // [rows][crop_w][3] results of the group
    const int nv = pitch >> 4;  // 16-byte vectors per LDS row image
    for (int idx = threadIdx.x; idx < rows * nv; idx += 256) {
        const int r = idx / nv, v = idx - r * nv;
        const size_t first = (size_t)g[G_SRC] + ((size_t)(row0 + r0 + r) * in_w) * 3 + lo;  // absolute byte in the packed buffer
        const size_t al = first & ~(size_t)15;
        const int nvec = (int)((first + (hi - lo) - al + 15) >> 4);  // never reads past the buffer: it is padded to 16
        if (v < nvec) reinterpret_cast<uint4*>(rows_lds + (size_t)r * pitch)[v] = reinterpret_cast<const uint4*>(a.src + al)[v];
    }
    __syncthreads();
    int rowoff[ROWS_PER_BLOCK];  // LDS byte of (row r, source byte `lo`)
#pragma unroll
    for (int r = 0; r < ROWS_PER_BLOCK; ++r) {
        const int rr = r < rows ? r : rows - 1;  // rows past the group re-read its last row (branch-free taps); not stored
        const size_t first = (size_t)g[G_SRC] + ((size_t)(row0 + r0 + rr) * in_w) * 3 + lo;
        rowoff[r] = rr * pitch + (int)(first & 15) - lo;
    }
    for (int j = threadIdx.x; j < a.crop_w; j += 256) {
        const int xmin = bd[j * 2], n = bd[j * 2 + 1];
        int acc[ROWS_PER_BLOCK][3];
#pragma unroll
        for (int r = 0; r < ROWS_PER_BLOCK; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 1 << (PRECISION_BITS - 1);
        // Four taps (12 contiguous bytes of each staged row) at a time: one unaligned ds_read_b64 + ds_read_b32 per row,
        // all rows in flight before the first use (a byte-by-byte tap loop spends 60 % of its cycles waiting on ~150
        // serialised LDS reads); taps past n carry a zero coefficient, their bytes may be anything.
        const unsigned abase = lds_byte_addr(rows_lds) + xmin * 3;
        for (int x0 = 0; x0 < n; x0 += 4) {
            int kx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) kx[u] = x0 + u < n ? coef[(size_t)(x0 + u) * a.crop_w + j] : 0;
            unsigned long long w8[ROWS_PER_BLOCK];
            unsigned w4[ROWS_PER_BLOCK];
#pragma unroll
            for (int r = 0; r < ROWS_PER_BLOCK; ++r) {
                const unsigned ad = abase + rowoff[r] + 3 * x0;
                asm volatile("ds_read_b64 %0, %2\n\tds_read_b32 %1, %2 offset:8" : "=&v"(w8[r]), "=&v"(w4[r]) : "v"(ad));  // early clobber: the address register must survive the first read
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < ROWS_PER_BLOCK; ++r) {
                // keep the asm outputs behind the wait
                asm volatile("" : "+v"(w8[r]), "+v"(w4[r]));
                const unsigned d0 = (unsigned)w8[r], d1 = (unsigned)(w8[r] >> 32), d2 = w4[r];
                acc[r][0] += __mul24((int)(d0 & 0xff), kx[0]) + __mul24((int)(d0 >> 24), kx[1]) +
                             __mul24((int)((d1 >> 16) & 0xff), kx[2]) + __mul24((int)((d2 >> 8) & 0xff), kx[3]);
                acc[r][1] += __mul24((int)((d0 >> 8) & 0xff), kx[0]) + __mul24((int)(d1 & 0xff), kx[1]) +
                             __mul24((int)(d1 >> 24), kx[2]) + __mul24((int)((d2 >> 16) & 0xff), kx[3]);
                acc[r][2] += __mul24((int)((d0 >> 16) & 0xff), kx[0]) + __mul24((int)((d1 >> 8) & 0xff), kx[1]) +
                             __mul24((int)(d2 & 0xff), kx[2]) + __mul24((int)(d2 >> 24), kx[3]);
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS_PER_BLOCK; ++r) {
            if (r < rows) {
                uint8_t* o = tile + (r * a.crop_w + j) * 3;
                o[0] = (uint8_t)clip8(acc[r][0]);
                o[1] = (uint8_t)clip8(acc[r][1]);
                o[2] = (uint8_t)clip8(acc[r][2]);
            }
        }
    }
    // the group's rows are one contiguous byte range of the intermediate image: copy the LDS tile out in 16-byte
    // vectors (three byte stores per lane, 3 bytes apart, cost 5x the whole pass)
    __syncthreads();
    const int stride = a.crop_w * 3, nbytes = rows * stride;
    uint8_t* o = dst + (size_t)r0 * stride;
    if (((g[G_INTER] + (int64_t)r0 * stride) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.inter) & 15) == 0) {
        for (int v = threadIdx.x; v < (nbytes >> 4); v += 256) reinterpret_cast<uint4*>(o)[v] = reinterpret_cast<const uint4*>(tile)[v];
        for (int e = (nbytes & ~15) + threadIdx.x; e < nbytes; e += 256) o[e] = tile[e];
    } else {
        for (int e = threadIdx.x; e < nbytes; e += 256) o[e] = tile[e];
    }
}

// u8 -> float of the rescale / normalise steps: 3 x 256 values, computed once per call exactly as transformers does
// (`float32(float64(u) * rescale_factor)`, then `(x - mean) / std` in float32).
__global__ __launch_bounds__(256) void pixel_lut_kernel(const ImgArgs a) {
    const int u = threadIdx.x, c = blockIdx.x;
    float v = (a.flags & MQ_IMAGE_RESCALE) ? (float)((double)u * a.rescale) : (float)u;
    if (a.flags & MQ_IMAGE_NORMALIZE) v = __fdiv_rn(__fsub_rn(v, a.mean[c]), a.stdv[c]);
    a.lut[c * 256 + u] = v;
}

// Vertical pass + rescale + normalise + HWC -> CHW.  blockIdx.y = image, blockIdx.x = 4 output rows.  Phase 1: a thread
// owns FOUR consecutive bytes of the interleaved (RGBRGB...) row, one aligned dword per tap row (taps and bounds are
// wave-uniform); the clipped bytes go to LDS.  Phase 2: the row leaves as three planar runs of floats, coalesced.
constexpr int COLS_ROWS = 4;  // output rows per workgroup of the column pass

__global__ __launch_bounds__(256) void resample_cols_kernel(const ImgArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t row_lds[];
    const int b = blockIdx.y, i0 = blockIdx.x * COLS_ROWS;
    const int64_t* g = a.geom + (size_t)b * G;
    const int stride = a.crop_w * 3, dw = (stride + 3) / 4;
    const int nr = min(COLS_ROWS, a.crop_h - i0);
    const int row0 = a.rowspan[b * 2 + 0], ksize = (int)g[G_KV];
    const uint8_t* inter = a.inter + g[G_INTER];
    const bool aligned = (stride & 3) == 0 && (reinterpret_cast<uintptr_t>(a.inter) & 3) == 0;
    for (int idx = threadIdx.x; idx < nr * dw; idx += 256) {
        const int ri = idx / dw, t = idx - ri * dw, i = i0 + ri;
        const int32_t* bd = a.bounds + (((size_t)b * 2 + 1) * a.crop_max + i) * 2;
        const int ymin = bd[0], n = bd[1];
        const int32_t* k = a.coefs + g[G_COEFV] + (size_t)i * ksize;
        const uint8_t* p = inter + (size_t)(ymin - row0) * stride + 4 * t;
        int s[4] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
        if (aligned) {
            for (int y = 0; y < n; ++y) {
                const int ky = k[y];
                const unsigned d = *reinterpret_cast<const unsigned*>(p + (size_t)y * stride);
                s[0] += __mul24((int)(d & 0xff), ky);
                s[1] += __mul24((int)((d >> 8) & 0xff), ky);
                s[2] += __mul24((int)((d >> 16) & 0xff), ky);
                s[3] += __mul24((int)(d >> 24), ky);
            }
        } else {  // odd crop widths: rows are not dword aligned
            const int nb = min(4, stride - 4 * t);
            for (int y = 0; y < n; ++y) {
                const int ky = k[y];
                for (int e = 0; e < nb; ++e) s[e] += __mul24((int)p[(size_t)y * stride + e], ky);
            }
        }
        reinterpret_cast<unsigned*>(row_lds)[ri * dw + t] = (unsigned)clip8(s[0]) | ((unsigned)clip8(s[1]) << 8) |
                                                            ((unsigned)clip8(s[2]) << 16) | ((unsigned)clip8(s[3]) << 24);
    }
    __syncthreads();
    const size_t plane = (size_t)a.crop_h * a.crop_w;
    float* o = a.out + (size_t)b * 3 * plane + (size_t)i0 * a.crop_w;
    for (int ri = 0; ri < nr; ++ri) {
        const uint8_t* row = row_lds + (size_t)ri * dw * 4;
        for (int j = threadIdx.x; j < a.crop_w; j += 256) {
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c * plane + (size_t)ri * a.crop_w + j] = a.lut[c * 256 + row[3 * j + c]];
        }
    }
}

inline size_t align_up(size_t v, size_t al) { return (v + al - 1) / al * al; }

}  // namespace

extern "C" {

int mq_image_plan(const int64_t* sizes_host, int n_images, int resize_mode, int size_h, int size_w, int crop_h, int crop_w,
                  int filter, int64_t* geom_host, int64_t* totals_host) {
    if (!sizes_host || !geom_host || !totals_host || n_images < 0 || crop_h < 1 || crop_w < 1) return MQ_EINVAL;
    if (filter != MQ_IMAGE_BICUBIC && filter != MQ_IMAGE_BILINEAR) return MQ_EUNSUPPORTED;
    if (resize_mode != MQ_IMAGE_RESIZE_NONE && resize_mode != MQ_IMAGE_RESIZE_SHORTEST && resize_mode != MQ_IMAGE_RESIZE_EXACT)
        return MQ_EINVAL;
    if (resize_mode != MQ_IMAGE_RESIZE_NONE && (size_h < 1 || (resize_mode == MQ_IMAGE_RESIZE_EXACT && size_w < 1))) return MQ_EINVAL;
    int64_t src = 0, coef = 0, inter = 0, max_rows = 0, max_w = 0;
    for (int b = 0; b < n_images; ++b) {
        const int64_t h = sizes_host[2 * b], w = sizes_host[2 * b + 1];
        if (h < 1 || w < 1 || h >= (1 << 24) || w >= (1 << 24)) return MQ_EINVAL;
        int64_t oh = h, ow = w;
        if (resize_mode == MQ_IMAGE_RESIZE_SHORTEST) {
            // transformers get_resize_output_image_size(default_to_square=False): int(size * long / short)
            const int64_t sh = w <= h ? w : h, lg = w <= h ? h : w;
            const int64_t nl = (int64_t)((double)(size_h * lg) / (double)sh);
            oh = w <= h ? nl : size_h;
            ow = w <= h ? size_h : nl;
        } else if (resize_mode == MQ_IMAGE_RESIZE_EXACT) {
            oh = size_h;
            ow = size_w;
        }
        if (oh < crop_h || ow < crop_w) return MQ_EUNSUPPORTED;  // HF would zero-pad: not provided
        if (oh >= (1 << 24) || ow >= (1 << 24)) return MQ_EINVAL;
        int64_t* g = geom_host + (size_t)b * G;
        g[G_SRC] = src;
        g[G_INH] = h; g[G_INW] = w; g[G_OUTH] = oh; g[G_OUTW] = ow;
        g[G_TOP] = (oh - crop_h) / 2;
        g[G_LEFT] = (ow - crop_w) / 2;
        g[G_KH] = resample_ksize((int)w, (int)ow, filter);
        g[G_KV] = resample_ksize((int)h, (int)oh, filter);
        g[G_COEFH] = coef; coef += (int64_t)crop_w * g[G_KH];
        g[G_COEFV] = coef; coef += (int64_t)crop_h * g[G_KV];
        g[G_INTER] = inter; inter += (int64_t)align_up((size_t)h * crop_w * 3, 16);
        src += (int64_t)align_up((size_t)h * w * 3, 16);
        if (h > max_rows) max_rows = h;
        if (w > max_w) max_w = w;
    }
    const int crop_max = crop_h > crop_w ? crop_h : crop_w;
    size_t ws = 0;
    ws += align_up((size_t)n_images * 2 * crop_max * 2 * sizeof(int32_t), 256);  // bounds
    ws += align_up((size_t)n_images * 2 * sizeof(int32_t), 256);                 // rowspan
    ws += align_up(3 * 256 * sizeof(float), 256);                                // u8 -> float table
    ws += align_up((size_t)coef * sizeof(int32_t), 256);                         // coefficients
    ws += align_up((size_t)inter, 256);                                          // horizontal-pass image
    totals_host[0] = src;
    totals_host[1] = (int64_t)ws;
    totals_host[2] = max_rows;
    totals_host[3] = coef;
    totals_host[4] = max_w;
    return MQ_OK;
}

int mq_image_preprocess_u8(const uint8_t* src_dev, const int64_t* geom_dev, int n_images, const int64_t* totals_host, int crop_h,
                           int crop_w, int filter, int flags, double rescale_factor, const float* mean3_host,
                           const float* std3_host, float* out_dev, void* ws_dev, size_t ws_bytes, void* stream) {
    if (n_images == 0) return MQ_OK;
    if (!src_dev || !geom_dev || !out_dev || !ws_dev || !totals_host || n_images < 0 || crop_h < 1 || crop_w < 1) return MQ_EINVAL;
    const int64_t max_rows = totals_host[2], coef_ints = totals_host[3], max_w = totals_host[4];
    if (max_rows < 1 || coef_ints < 1 || max_w < 1 || (reinterpret_cast<uintptr_t>(src_dev) & 15)) return MQ_EINVAL;
    // LDS image of one source row: its crop-window byte span, shifted by its offset inside a 16-byte line
    const int64_t pitch = (max_w * 3 + 32 + 15) / 16 * 16;
    if (pitch > 64 * 1024) return MQ_EUNSUPPORTED;  // images wider than ~21 k pixels
    if (filter != MQ_IMAGE_BICUBIC && filter != MQ_IMAGE_BILINEAR) return MQ_EUNSUPPORTED;
    if ((flags & MQ_IMAGE_NORMALIZE) && (!mean3_host || !std3_host)) return MQ_EINVAL;
    ImgArgs a;
    a.src = src_dev;
    a.geom = geom_dev;
    a.B = n_images; a.crop_h = crop_h; a.crop_w = crop_w; a.crop_max = crop_h > crop_w ? crop_h : crop_w;
    a.filter = filter; a.flags = flags; a.rescale = rescale_factor;
    for (int c = 0; c < 3; ++c) {
        a.mean[c] = mean3_host ? mean3_host[c] : 0.f;
        a.stdv[c] = std3_host ? std3_host[c] : 1.f;
    }
    char* p = static_cast<char*>(ws_dev);
    size_t off = 0;
    a.bounds = reinterpret_cast<int32_t*>(p + off); off += align_up((size_t)n_images * 2 * a.crop_max * 2 * sizeof(int32_t), 256);
    a.rowspan = reinterpret_cast<int32_t*>(p + off); off += align_up((size_t)n_images * 2 * sizeof(int32_t), 256);
    a.lut = reinterpret_cast<float*>(p + off); off += align_up(3 * 256 * sizeof(float), 256);
    a.coefs = reinterpret_cast<int32_t*>(p + off); off += align_up((size_t)coef_ints * sizeof(int32_t), 256);
    a.inter = reinterpret_cast<uint8_t*>(p + off);
    if (off > ws_bytes) return MQ_EWORKSPACE;
    a.out = out_dev;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(resample_coeffs_kernel, dim3((a.crop_max + 255) / 256, n_images, 2), dim3(256), 0, st, a);
    a.row_pitch = (int)pitch;
    {   // rows per group: staged rows + result tile within the 64 KB a kernel gets without opting in to more
        const int64_t per_row = pitch + (int64_t)crop_w * 3 + 16;
        if (per_row > 64 * 1024) return MQ_EUNSUPPORTED;
        a.rpb = (int)(64 * 1024 / per_row) < ROWS_PER_BLOCK ? (int)(64 * 1024 / per_row) : ROWS_PER_BLOCK;
    }
    const size_t tile_bytes = align_up((size_t)a.rpb * crop_w * 3, 16);
    hipLaunchKernelGGL(pixel_lut_kernel, dim3(3), dim3(256), 0, st, a);
    hipLaunchKernelGGL(resample_rows_kernel, dim3((unsigned)((max_rows + a.rpb - 1) / a.rpb), n_images), dim3(256),
                       (size_t)a.rpb * pitch + tile_bytes, st, a);
    hipLaunchKernelGGL(resample_cols_kernel, dim3((crop_h + COLS_ROWS - 1) / COLS_ROWS, n_images), dim3(256),
                       (size_t)COLS_ROWS * align_up((size_t)crop_w * 3 + 4, 16), st, a);
    IMG_HIP(hipGetLastError());
    return MQ_OK;
}

}  // extern "C"
