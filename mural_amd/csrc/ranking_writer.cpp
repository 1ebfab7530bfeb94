// ranking_writer.cpp — host-side text output of the retrieval CLI.
//
// Replaces the per-hit Python loop of write_ranking
// (src/tevatron/faiss_retriever/__main__.py:32-38):
//     for each query: sorted(zip(scores, ids), key=score, reverse=True)   (stable)
//                     f.write(f'{qid}\t{idx}\t{s}\n')
// `s` is a numpy.float32, whose __format__/__str__ go through Python's float:
// the float32 is widened to double and printed with repr(float) — the shortest
// decimal string that round-trips the DOUBLE (CPython float_repr_style 'short',
// format code 'r': exponent form when the decimal exponent is < -4 or >= 16,
// at least two exponent digits, and a trailing ".0" on integral values).
#include "../../include/ipx.h"

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <functional>
#include <thread>
#include <vector>

namespace {

// repr(float(x)) for a finite or non-finite double; returns the number of chars written.
int py_float_repr(double v, char* out) {
    if (std::isnan(v)) { memcpy(out, "nan", 3); return 3; }
    if (std::isinf(v)) {
        if (v < 0) { memcpy(out, "-inf", 4); return 4; }
        memcpy(out, "inf", 3);
        return 3;
    }
    char* p = out;
    if (std::signbit(v)) { *p++ = '-'; v = -v; }
    if (v == 0.0) { memcpy(p, "0.0", 3); return (int)(p - out) + 3; }
    // shortest round-trip digits, scientific: d[.ddd]e[+-]XX
    char sci[40];
    auto res = std::to_chars(sci, sci + sizeof(sci), v, std::chars_format::scientific);
    char digits[24];
    int nd = 0;
    const char* c = sci;
    for (; c < res.ptr && *c != 'e'; ++c)
        if (*c != '.') digits[nd++] = *c;
    int e10 = 0;
    {
        ++c;  // 'e'
        bool neg = false;
        if (*c == '+') ++c;
        else if (*c == '-') { neg = true; ++c; }
        for (; c < res.ptr; ++c) e10 = e10 * 10 + (*c - '0');
        if (neg) e10 = -e10;
    }
    const int decpt = e10 + 1;  // value = 0.d1d2... * 10^decpt
    if (decpt <= -4 || decpt > 16) {
        *p++ = digits[0];
        if (nd > 1) {
            *p++ = '.';
            memcpy(p, digits + 1, nd - 1);
            p += nd - 1;
        }
        *p++ = 'e';
        int e = decpt - 1;
        if (e < 0) { *p++ = '-'; e = -e; } else { *p++ = '+'; }
        char eb[8];
        int ne = 0;
        do { eb[ne++] = (char)('0' + e % 10); e /= 10; } while (e);
        if (ne < 2) eb[ne++] = '0';
        while (ne) *p++ = eb[--ne];
    } else if (decpt <= 0) {
        *p++ = '0';
        *p++ = '.';
        for (int i = 0; i < -decpt; ++i) *p++ = '0';
        memcpy(p, digits, nd);
        p += nd;
    } else if (decpt >= nd) {
        memcpy(p, digits, nd);
        p += nd;
        for (int i = nd; i < decpt; ++i) *p++ = '0';
        *p++ = '.';
        *p++ = '0';
    } else {
        memcpy(p, digits, decpt);
        p += decpt;
        *p++ = '.';
        memcpy(p, digits + decpt, nd - decpt);
        p += nd - decpt;
    }
    return (int)(p - out);
}

}  // namespace

// defined in ipx.hip; lets this file report through ipx_last_error()
extern "C" void ipx_set_error_(const char* msg);

extern "C" int ipx_format_score(float x, char* buf, size_t buflen) {
    char tmp[48];
    const int n = py_float_repr((double)x, tmp);
    if (!buf || buflen < (size_t)n + 1) return -1;
    memcpy(buf, tmp, n);
    buf[n] = 0;
    return n;
}

namespace {

// one id from a numpy '<U' cell (UCS-4 code points, zero padded to `width`) as UTF-8
inline char* put_ucs4(char* p, const uint32_t* s, int width) {
    for (int i = 0; i < width && s[i]; ++i) {
        const uint32_t c = s[i];
        if (c < 0x80) *p++ = (char)c;
        else if (c < 0x800) { *p++ = (char)(0xC0 | (c >> 6)); *p++ = (char)(0x80 | (c & 0x3F)); }
        else if (c < 0x10000) {
            *p++ = (char)(0xE0 | (c >> 12)); *p++ = (char)(0x80 | ((c >> 6) & 0x3F)); *p++ = (char)(0x80 | (c & 0x3F));
        } else {
            *p++ = (char)(0xF0 | (c >> 18)); *p++ = (char)(0x80 | ((c >> 12) & 0x3F));
            *p++ = (char)(0x80 | ((c >> 6) & 0x3F)); *p++ = (char)(0x80 | (c & 0x3F));
        }
    }
    return p;
}

inline char* put_i64(char* p, int64_t v) {
    auto r = std::to_chars(p, p + 24, v);
    return r.ptr;
}

}  // namespace

extern "C" int ipx_write_ranking(const char* path, const uint32_t* qids, int qid_width, const uint32_t* docids,
                                 int docid_width, const int64_t* docids_i64, const float* scores, int64_t nq,
                                 int k) {
    if (!path || nq < 0 || k < 0 || qid_width < 0 || docid_width < 0 ||
        (nq > 0 && (!qids || (k > 0 && ((!docids && !docids_i64) || !scores))))) {
        ipx_set_error_("write_ranking: bad argument");
        return IPX_ERR_ARG;
    }
    FILE* f = fopen(path, "w");
    if (!f) {
        ipx_set_error_((std::string("write_ranking: cannot open ") + path).c_str());
        return IPX_ERR_IO;
    }
    const size_t line_max = 4 * (size_t)qid_width + (docids ? 4 * (size_t)docid_width : 24) + 48;
    // Queries are formatted in blocks by a few threads (sorting + shortest-round-trip float printing is
    // the cost: 7M lines at depth 1000), and the blocks are written in order by this thread.
    const int64_t block = std::max<int64_t>(1, std::min<int64_t>(256, (int64_t)((4u << 20) / (line_max * (size_t)(k > 0 ? k : 1) + 1))));
    int nthreads = (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    if (const char* e = std::getenv("IPX_WRITER_THREADS")) nthreads = std::max(1, std::min(64, std::atoi(e)));
    auto format_block = [&](int64_t q0, int64_t q1, std::vector<char>& out) {
        out.resize((size_t)(q1 - q0) * line_max * (size_t)(k > 0 ? k : 1) + 64);
        std::vector<int> order(k);
        char* p = out.data();
        for (int64_t i = q0; i < q1; ++i) {
            const float* s = scores + i * k;
            std::iota(order.begin(), order.end(), 0);
            // stable, descending by score only (ties keep input order, as Python's sorted does)
            std::stable_sort(order.begin(), order.end(), [s](int a, int b) {
                const float x = std::isnan(s[a]) ? -INFINITY : s[a];
                const float y = std::isnan(s[b]) ? -INFINITY : s[b];
                return x > y;
            });
            for (int j = 0; j < k; ++j) {
                const int h = order[j];
                p = put_ucs4(p, qids + i * qid_width, qid_width);
                *p++ = '\t';
                if (docids) p = put_ucs4(p, docids + ((size_t)i * k + h) * docid_width, docid_width);
                else p = put_i64(p, docids_i64[(size_t)i * k + h]);
                *p++ = '\t';
                p += py_float_repr((double)s[h], p);
                *p++ = '\n';
            }
        }
        out.resize((size_t)(p - out.data()));
    };
    std::vector<std::vector<char>> bufs((size_t)nthreads);
    bool ok = true;
    for (int64_t q0 = 0; q0 < nq && ok; q0 += block * nthreads) {
        std::vector<std::thread> th;
        int used_bufs = 0;
        for (int t = 0; t < nthreads; ++t) {
            const int64_t b0 = q0 + t * block, b1 = std::min(nq, b0 + block);
            if (b0 >= nq) break;
            ++used_bufs;
            if (nthreads == 1 || nq <= block) format_block(b0, b1, bufs[(size_t)t]);
            else th.emplace_back(format_block, b0, b1, std::ref(bufs[(size_t)t]));
        }
        for (auto& t : th) t.join();
        for (int t = 0; t < used_bufs && ok; ++t)
            ok = bufs[(size_t)t].empty() || fwrite(bufs[(size_t)t].data(), 1, bufs[(size_t)t].size(), f) == bufs[(size_t)t].size();
    }
    if (fclose(f) != 0) ok = false;
    if (!ok) {
        ipx_set_error_((std::string("write_ranking: write failed on ") + path).c_str());
        return IPX_ERR_IO;
    }
    return 0;
}
