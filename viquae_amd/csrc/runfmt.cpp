// Host side of SURVEY 8 row f1: a search job's result arrays -> the text of the run file, without the per-hit Python objects.
//
// The reference ends `dataset_search` with `run.save(metric_save_path / f"{index_name}.json")`
// (meerqat/ir/search.py:485-498) over `runs[index_name][q_id][str(doc)] = score` dicts filled by a Python triple loop
// (:413-440): at 16,384 questions x 100 hits that is 1.6 M dict inserts and 1.6 M float reprs.  This build keeps a job's
// [nq, k] result arrays as arrays to the end (viquae_amd/ir/runs.py) and formats the JSON text from them here, on the host
// cores, byte for byte what `json.dump({q: {str(doc): float(score)}})` writes: `", "` / `": "` separators, Python's
// `float.__repr__` (shortest digits that round-trip the double; exponent form when the decimal exponent is < -4 or >= 16;
// "NaN" / "Infinity" / "-Infinity" as json.dump spells them).  No device code in this file.
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/meerqat_hip.h"

namespace {

// repr(float) of CPython (Python/pystrtod.c format_float_short, mode 'r'): digits d1 d2 ... dn and a decimal point position
// `decpt` (value = 0.d1...dn x 10^decpt); exponent notation iff decpt <= -4 or decpt > 16; ".0" appended to integers.
inline char* py_float_repr(double v, char* p) {
    if (std::isnan(v)) { std::memcpy(p, "NaN", 3); return p + 3; }
    if (std::isinf(v)) {
        if (v < 0) *p++ = '-';
        std::memcpy(p, "Infinity", 8);
        return p + 8;
    }
    char buf[48];
    const auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::scientific);  // [-]d[.ddd]e[+-]XX, shortest
    const char* s = buf;
    if (*s == '-') *p++ = *s++;
    char dg[32];
    int nd = 0;
    dg[nd++] = *s++;
    if (*s == '.') {
        ++s;
        while (*s != 'e') dg[nd++] = *s++;
    }
    ++s;  // 'e'
    const bool eneg = *s == '-';
    ++s;
    int e = 0;
    while (s < r.ptr) e = e * 10 + (*s++ - '0');
    const int decpt = (eneg ? -e : e) + 1;
    if (decpt <= -4 || decpt > 16) {
        *p++ = dg[0];
        if (nd > 1) {
            *p++ = '.';
            std::memcpy(p, dg + 1, nd - 1);
            p += nd - 1;
        }
        *p++ = 'e';
        int ex = decpt - 1;
        if (ex < 0) { *p++ = '-'; ex = -ex; } else { *p++ = '+'; }
        if (ex >= 100) { *p++ = char('0' + ex / 100); ex %= 100; }
        *p++ = char('0' + ex / 10);
        *p++ = char('0' + ex % 10);
    } else if (decpt <= 0) {
        *p++ = '0';
        *p++ = '.';
        for (int i = 0; i < -decpt; ++i) *p++ = '0';
        std::memcpy(p, dg, nd);
        p += nd;
    } else if (decpt >= nd) {
        std::memcpy(p, dg, nd);
        p += nd;
        for (int i = nd; i < decpt; ++i) *p++ = '0';
        *p++ = '.';
        *p++ = '0';
    } else {
        std::memcpy(p, dg, decpt);
        p += decpt;
        *p++ = '.';
        std::memcpy(p, dg + decpt, nd - decpt);
        p += nd - decpt;
    }
    return p;
}

inline char* put_i64(int64_t v, char* p) {
    const auto r = std::to_chars(p, p + 24, v);
    return r.ptr;
}

struct RunArgs {
    const char* qid_json;
    const int64_t* qid_off;
    const int64_t* ids;
    const void* scores;
    int scores_f64;
    int64_t stride;
    const int32_t* counts;
};

// worst case of one entry: "<19 digits>": <24 chars>,<space> = 50 bytes; of a row without its entries: <qid>: {},<space> = 6 bytes
constexpr int64_t ENTRY_MAX = 52, ROW_MAX = 8;

char* format_rows(const RunArgs& a, int64_t q0, int64_t q1, char* p) {
    // "<qid>": {"<id>": <score>, ...}  -- rows joined by ", "
    for (int64_t q = q0; q < q1; ++q) {
        if (q > q0) { *p++ = ','; *p++ = ' '; }
        const int64_t ql = a.qid_off[q + 1] - a.qid_off[q];
        std::memcpy(p, a.qid_json + a.qid_off[q], (size_t)ql);
        p += ql;
        *p++ = ':'; *p++ = ' '; *p++ = '{';
        const int64_t n = a.counts ? a.counts[q] : a.stride;
        const int64_t* id = a.ids + q * a.stride;
        bool first = true;
        for (int64_t j = 0; j < n && j < a.stride; ++j) {
            if (id[j] < 0) break;  // unfilled slots end a row (FAISS's -1 padding)
            if (!first) { *p++ = ','; *p++ = ' '; }
            first = false;
            *p++ = '"';
            p = put_i64(id[j], p);
            *p++ = '"'; *p++ = ':'; *p++ = ' ';
            const double v = a.scores_f64 ? static_cast<const double*>(a.scores)[q * a.stride + j]
                                          : (double)static_cast<const float*>(a.scores)[q * a.stride + j];
            p = py_float_repr(v, p);
        }
        *p++ = '}';
    }
    return p;
}

}  // namespace

extern "C" {

int64_t mq_format_run_json(const char* qid_json, const int64_t* qid_off, int64_t nq, const int64_t* ids, const void* scores,
                           int scores_f64, int64_t stride, const int32_t* counts, char* out, int64_t out_cap, int n_threads,
                           int64_t* parts) {
    if (nq < 0 || stride < 0 || !parts || (nq > 0 && (!qid_json || !qid_off || (stride > 0 && (!ids || !scores))))) return MQ_EINVAL;
    if (nq == 0) return 0;
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > MQ_RUN_JSON_MAX_PARTS) nt = MQ_RUN_JSON_MAX_PARTS;
    if (nq * (stride > 0 ? stride : 1) < (1 << 14)) nt = 1;  // a thread is not worth starting for a few thousand entries
    if ((int64_t)nt > nq) nt = (int)nq;
    const int64_t need = (qid_off[nq] - qid_off[0]) + nq * (ROW_MAX + stride * ENTRY_MAX);
    if (!out || out_cap < need) return -need - 16;  // retry with -(rc + 16) bytes
    const RunArgs a{qid_json, qid_off, ids, scores, scores_f64, stride, counts};
    // every part formats its rows into its own worst-case region of `out`: no second buffer, no compaction -- the caller writes
    // the parts one after the other, ", " between them
    auto work = [&](int t) {
        const int64_t q0 = nq * t / nt, q1 = nq * (t + 1) / nt;
        const int64_t begin = (qid_off[q0] - qid_off[0]) + q0 * (ROW_MAX + stride * ENTRY_MAX);
        char* const end = format_rows(a, q0, q1, out + begin);
        parts[2 * t] = begin;
        parts[2 * t + 1] = (int64_t)(end - (out + begin));
    };
    if (nt == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
    }
    return nt;
}

}  // extern "C"
