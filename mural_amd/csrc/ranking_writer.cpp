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
#include <atomic>
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

namespace {

// Shared body of the two writers: `put_doc(p, i, h)` appends the doc id of hit h of query i; `doc_max` bounds its
// UTF-8 length. Never throws: an allocation failure in a formatting thread becomes IPX_ERR_NOMEM.
template <class PutDoc>
int write_ranking_impl(const char* path, const uint32_t* qids, int qid_width, size_t doc_max, PutDoc put_doc,
                       const float* scores, int64_t nq, int k) {
    FILE* f = fopen(path, "w");
    if (!f) {
        ipx_set_error_((std::string("write_ranking: cannot open ") + path).c_str());
        return IPX_ERR_IO;
    }
    const size_t line_max = 4 * (size_t)qid_width + doc_max + 48;
    // Queries are formatted in blocks by a few threads (sorting + shortest-round-trip float printing is
    // the cost: 7M lines at depth 1000), and the blocks are written in order by this thread.
    const int64_t block = std::max<int64_t>(1, std::min<int64_t>(256, (int64_t)((4u << 20) / (line_max * (size_t)(k > 0 ? k : 1) + 1))));
    int nthreads = (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    if (const char* e = std::getenv("IPX_WRITER_THREADS")) nthreads = std::max(1, std::min(64, std::atoi(e)));
    std::atomic<bool> nomem{false};
    auto format_block = [&](int64_t q0, int64_t q1, std::vector<char>& out) {
        try {
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
                    p = put_doc(p, i, h);
                    *p++ = '\t';
                    p += py_float_repr((double)s[h], p);
                    *p++ = '\n';
                }
            }
            out.resize((size_t)(p - out.data()));
        } catch (...) {             // bad_alloc: reported, the process lives on
            out.clear();
            nomem = true;
        }
    };
    bool ok = true;
    try {
        std::vector<std::vector<char>> bufs((size_t)nthreads);
        for (int64_t q0 = 0; q0 < nq && ok && !nomem; q0 += block * nthreads) {
            std::vector<std::thread> th;
            int used_bufs = 0;
            for (int t = 0; t < nthreads; ++t) {
                const int64_t b0 = q0 + t * block, b1 = std::min(nq, b0 + block);
                if (b0 >= nq) break;
                ++used_bufs;
                if (nthreads == 1 || nq <= block) format_block(b0, b1, bufs[(size_t)t]);
                else {
                    try {
                        th.emplace_back(format_block, b0, b1, std::ref(bufs[(size_t)t]));
                    } catch (...) {                 // no thread to be had: format here
                        format_block(b0, b1, bufs[(size_t)t]);
                    }
                }
            }
            for (auto& t : th) t.join();
            for (int t = 0; t < used_bufs && ok && !nomem; ++t)
                ok = bufs[(size_t)t].empty() || fwrite(bufs[(size_t)t].data(), 1, bufs[(size_t)t].size(), f) == bufs[(size_t)t].size();
        }
    } catch (...) {
        nomem = true;
    }
    if (fclose(f) != 0) ok = false;
    if (nomem) {
        ipx_set_error_("write_ranking: out of host memory while formatting");
        return IPX_ERR_NOMEM;
    }
    if (!ok) {
        ipx_set_error_((std::string("write_ranking: write failed on ") + path).c_str());
        return IPX_ERR_IO;
    }
    return 0;
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
    if (docids)
        return write_ranking_impl(path, qids, qid_width, 4 * (size_t)docid_width,
                                  [=](char* p, int64_t i, int h) { return put_ucs4(p, docids + ((size_t)i * k + h) * docid_width, docid_width); },
                                  scores, nq, k);
    return write_ranking_impl(path, qids, qid_width, 24,
                              [=](char* p, int64_t i, int h) { return put_i64(p, docids_i64[(size_t)i * k + h]); }, scores, nq, k);
}

extern "C" int ipx_write_ranking_lookup(const char* path, const uint32_t* qids, int qid_width, const char* doc_blob,
                                        const int64_t* doc_offsets, int sep, int64_t n_docs, const int64_t* rows,
                                        const float* scores, int64_t nq, int k) {
    if (!path || nq < 0 || k < 0 || qid_width < 0 || n_docs < 0 || sep < 0 ||
        (nq > 0 && (!qids || (k > 0 && (!rows || !scores || !doc_offsets || (!doc_blob && n_docs > 0)))))) {
        ipx_set_error_("write_ranking_lookup: bad argument");
        return IPX_ERR_ARG;
    }
    // every hit must name a row of the table (negative rows count from the end, as list indexing in the
    // reference does: __main__.py:27); the longest id bounds a line
    size_t doc_max = 0;
    for (int64_t r = 0; r < n_docs; ++r) {
        if (doc_offsets[r + 1] - sep < doc_offsets[r]) { ipx_set_error_("write_ranking_lookup: offsets not ascending"); return IPX_ERR_ARG; }
        doc_max = std::max(doc_max, (size_t)(doc_offsets[r + 1] - sep - doc_offsets[r]));
    }
    for (int64_t e = 0; e < nq * (int64_t)k; ++e)
        if (rows[e] >= n_docs || rows[e] < -n_docs) { ipx_set_error_("write_ranking_lookup: list index out of range"); return IPX_ERR_ARG; }
    return write_ranking_impl(path, qids, qid_width, doc_max,
                              [=](char* p, int64_t i, int h) {
                                  int64_t r = rows[(size_t)i * k + h];
                                  if (r < 0) r += n_docs;
                                  const size_t len = (size_t)(doc_offsets[r + 1] - sep - doc_offsets[r]);
                                  memcpy(p, doc_blob + doc_offsets[r], len);
                                  return p + len;
                              },
                              scores, nq, k);
}
