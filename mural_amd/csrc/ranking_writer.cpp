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
#include <cstring>
#include <numeric>
#include <string>
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

extern "C" int ipx_write_ranking(const char* path, const char* const* qids, const char* const* docids,
                                 const float* scores, int64_t nq, int k) {
    if (!path || nq < 0 || k < 0 || (nq > 0 && (!qids || (k > 0 && (!docids || !scores))))) {
        ipx_set_error_("write_ranking: bad argument");
        return IPX_ERR_ARG;
    }
    FILE* f = fopen(path, "w");
    if (!f) {
        ipx_set_error_((std::string("write_ranking: cannot open ") + path).c_str());
        return IPX_ERR_IO;
    }
    std::vector<char> buf;
    buf.reserve(1 << 22);
    std::vector<int> order(k);
    bool ok = true;
    for (int64_t i = 0; i < nq && ok; ++i) {
        const float* s = scores + i * k;
        const char* const* ids = docids + i * k;
        std::iota(order.begin(), order.end(), 0);
        // stable, descending by score only (ties keep input order, as Python's sorted does)
        std::stable_sort(order.begin(), order.end(), [s](int a, int b) {
            const float x = std::isnan(s[a]) ? -INFINITY : s[a];
            const float y = std::isnan(s[b]) ? -INFINITY : s[b];
            return x > y;
        });
        const size_t qlen = strlen(qids[i]);
        for (int j = 0; j < k; ++j) {
            const int h = order[j];
            const size_t dlen = strlen(ids[h]);
            const size_t need = qlen + dlen + 48;
            const size_t at = buf.size();
            buf.resize(at + need);
            char* p = buf.data() + at;
            memcpy(p, qids[i], qlen); p += qlen;
            *p++ = '\t';
            memcpy(p, ids[h], dlen); p += dlen;
            *p++ = '\t';
            p += py_float_repr((double)s[h], p);
            *p++ = '\n';
            buf.resize(p - buf.data());
        }
        if (buf.size() > (1u << 22) - 4096) {
            ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
            buf.clear();
        }
    }
    if (ok && !buf.empty()) ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    if (fclose(f) != 0) ok = false;
    if (!ok) {
        ipx_set_error_((std::string("write_ranking: write failed on ") + path).c_str());
        return IPX_ERR_IO;
    }
    return 0;
}
