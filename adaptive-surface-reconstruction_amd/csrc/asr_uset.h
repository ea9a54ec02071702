// asr_uset.h -- iteration order of a libstdc++ std::unordered_set<size_t>.
//
// The reference starts the cyclic ordering of the dual cells around an edge from the LAST element
// of an unordered_set iteration (cpp/lib/contouring.cpp:250-256), so its triangle corner order is a
// function of libstdc++'s hashtable: identity hash, bucket = x % bucket_count, 13 buckets after
// the first insert and 29 from the 14th on (GCC 11 _Prime_rehash_policy), new nodes go to the head
// of their bucket's group or, for an empty bucket, to the head of the whole list
// (bits/hashtable.h _M_insert_bucket_begin / _M_rehash_aux).  This header replays exactly that on
// a small array; tests/test_oracle_mesh.py checks it against the real container.
#pragma once
#include <cstdint>

#define ASR_USET_CAP 29  // elements supported (one rehash, 13 -> 29 buckets)

#if defined(__HIPCC__)
#define ASR_HD __host__ __device__
#else
#define ASR_HD
#endif

ASR_HD static inline void asr_uset_put(uint32_t* list, int& m, uint32_t x, uint32_t buckets) {
    const uint32_t b = x % buckets;
    int pos = 0;
    for (int j = 0; j < m; ++j)
        if (list[j] % buckets == b) {
            pos = j;
            break;
        }
    for (int j = m; j > pos; --j) list[j] = list[j - 1];
    list[pos] = x;
    ++m;
}

// xs: the inserted values in insertion order (distinct), n <= ASR_USET_CAP; out: iteration order
ASR_HD static inline void asr_uset_order(const uint32_t* xs, int n, uint32_t* out) {
    int m = 0;
    uint32_t buckets = 13;
    for (int i = 0; i < n; ++i) {
        if (i == 13) {  // rehash before the 14th insert: replay the list into 29 buckets
            uint32_t old[13];
            for (int j = 0; j < 13; ++j) old[j] = out[j];
            m = 0;
            buckets = 29;
            for (int j = 0; j < 13; ++j) asr_uset_put(out, m, old[j], buckets);
        }
        asr_uset_put(out, m, xs[i], buckets);
    }
}
