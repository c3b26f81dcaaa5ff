// ria_amd/csrc/sort_exact.hpp — libstdc++ std::sort(first, last, comp), restated so that it runs on one
// GPU lane as well as on the host, and produces the SAME permutation as the library does.
//
// Why it matters: the CRC recovery of v2::decodeFixedFrame sorts its "suspect" bits with std::sort and a
// comparator on |LLR| only (frame_v2.cpp:1717-1718).  std::sort is not stable, so the order of equal
// keys — which decides the suspects that are tried — is whatever GCC's introsort leaves behind:
//   __introsort_loop (median-of-3 to *first, unguarded Hoare partition, recurse right / loop left,
//   heapsort when the depth budget 2*floor(log2 n) is spent) over ranges longer than 16, then
//   __final_insertion_sort (guarded insertion on the first 16, unguarded linear inserts after).
// This is the published algorithm of libstdc++ (bits/stl_algo.h, bits/stl_heap.h), not reference code.
//
// sort_exact_prefix() is the same algorithm restricted to what can influence positions [0, want):
// partitions that lie entirely at or beyond `want + 16` never exchange elements with the prefix, and
// the final insertion pass never moves an element across a partition boundary (strict comparator), so
// the prefix comes out identical while the work drops from O(n log n) to O(n).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RIA_SORT_HD __host__ __device__
#else
#define RIA_SORT_HD
#endif

namespace ria {

struct Suspect { int frame_bit; float abs_llr; };
RIA_SORT_HD inline bool suspect_lt(const Suspect& a, const Suspect& b) { return a.abs_llr < b.abs_llr; }

namespace sortx {

template <class T> RIA_SORT_HD inline void swp(T& a, T& b) { T t = a; a = b; b = t; }

// ---- bits/stl_heap.h
template <class T, class C>
RIA_SORT_HD inline void push_heap(T* first, int hole, int top, T value, C lt) {
    int parent = (hole - 1) / 2;
    while (hole > top && lt(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
template <class T, class C>
RIA_SORT_HD inline void adjust_heap(T* first, int hole, int len, T value, C lt) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (lt(first[child], first[child - 1])) --child;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    push_heap(first, hole, top, value, lt);
}
template <class T, class C>
RIA_SORT_HD inline void heap_sort(T* first, T* last, C lt) {   // __partial_sort(first, last, last)
    const int len = static_cast<int>(last - first);
    if (len >= 2)
        for (int parent = (len - 2) / 2;; --parent) {
            T v = first[parent];
            adjust_heap(first, parent, len, v, lt);
            if (parent == 0) break;
        }
    while (last - first > 1) {
        --last;
        T v = *last;
        *last = *first;
        adjust_heap(first, 0, static_cast<int>(last - first), v, lt);
    }
}

// ---- bits/stl_algo.h
template <class T, class C>
RIA_SORT_HD inline void unguarded_linear_insert(T* last, C lt) {
    T val = *last;
    T* next = last - 1;
    while (lt(val, *next)) { *last = *next; last = next; --next; }
    *last = val;
}
template <class T, class C>
RIA_SORT_HD inline void insertion_sort(T* first, T* last, C lt) {
    if (first == last) return;
    for (T* i = first + 1; i != last; ++i) {
        if (lt(*i, *first)) {
            T v = *i;
            for (T* p = i; p != first; --p) *p = *(p - 1);
            *first = v;
        } else {
            unguarded_linear_insert(i, lt);
        }
    }
}
template <class T, class C>
RIA_SORT_HD inline T* partition_pivot(T* first, T* last, C lt) {   // __unguarded_partition_pivot
    T *mid = first + (last - first) / 2, *a = first + 1, *b = mid, *c = last - 1;
    if (lt(*a, *b)) {
        if (lt(*b, *c)) swp(*first, *b); else if (lt(*a, *c)) swp(*first, *c); else swp(*first, *a);
    } else {
        if (lt(*a, *c)) swp(*first, *a); else if (lt(*b, *c)) swp(*first, *c); else swp(*first, *b);
    }
    T *lo = first + 1, *hi = last;
    for (;;) {
        while (lt(*lo, *first)) ++lo;
        --hi;
        while (lt(*first, *hi)) --hi;
        if (!(lo < hi)) return lo;
        swp(*lo, *hi);
        ++lo;
    }
}

}  // namespace sortx

// Leaves v[0 .. min(n, want)) exactly as std::sort(v, v + n, lt) would.  `stack` holds 3 ints per pending
// range (the recursion of __introsort_loop): 3 * 64 ints are plenty (depth <= 2*log2 n).
// depth_budget < 0: the library's 2*floor(log2 n); tests pass 0 to force the heapsort branch.
template <class T, class C>
RIA_SORT_HD inline void sort_exact_prefix(T* v, int n, int want, int* stack, C lt, int depth_budget = -1) {
    if (n <= 0) return;
    const int limit = (want >= n) ? n : want + 16;   // positions at or beyond this never reach the prefix
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) ++lg;
    int sp = 0;
    stack[0] = 0; stack[1] = n; stack[2] = depth_budget < 0 ? 2 * lg : depth_budget;
    sp = 1;
    while (sp > 0) {
        --sp;
        int first = stack[3 * sp], last = stack[3 * sp + 1], depth = stack[3 * sp + 2];
        while (last - first > 16) {
            if (first >= limit) break;
            if (depth == 0) { sortx::heap_sort(v + first, v + last, lt); break; }
            --depth;
            const int cut = static_cast<int>(sortx::partition_pivot(v + first, v + last, lt) - v);
            if (cut < limit) { stack[3 * sp] = cut; stack[3 * sp + 1] = last; stack[3 * sp + 2] = depth; ++sp; }
            last = cut;
        }
    }
    const int fin = (limit < n) ? limit : n;
    if (n > 16) {
        sortx::insertion_sort(v, v + 16, lt);
        for (T* i = v + 16; i != v + fin; ++i) sortx::unguarded_linear_insert(i, lt);
    } else {
        sortx::insertion_sort(v, v + n, lt);
    }
}

// ---- the same result computed in a data-parallel way ------------------------------------------------------------
// __unguarded_partition is a two-pointer loop, but what it does is fixed by two lists: A = the positions in
// (first, last) whose key is NOT below the pivot, ascending, and B = the positions whose key is NOT above it, descending.
// Swap k of the loop exchanges A[k] and B[k]; the loop stops at the first k with A[k] >= B[k] and returns
// min(A[k], B[k-1]) (the left scan stops on the element the previous swap parked on the right, the right scan on the
// one it parked on the left).  Counting and list building are prefix sums: on the GPU a wave does a whole pass with
// ballots (recovery_kernels.hip.h); this host form states the rule and is checked against std::sort.
// After the partitions, __final_insertion_sort is a stable sort (strict comparator, elements only pass strictly
// greater ones) of the first min(n, want + 16) positions: a rank computation.
template <class T, class C>
RIA_SORT_HD inline int partition_lists(T* v, int first, int last, C lt, int* alist, int* blist) {
    T *f = v + first, *mid = v + first + (last - first) / 2, *a = v + first + 1, *b = mid, *c = v + last - 1;
    if (lt(*a, *b)) {
        if (lt(*b, *c)) sortx::swp(*f, *b); else if (lt(*a, *c)) sortx::swp(*f, *c); else sortx::swp(*f, *a);
    } else {
        if (lt(*a, *c)) sortx::swp(*f, *a); else if (lt(*b, *c)) sortx::swp(*f, *c); else sortx::swp(*f, *b);
    }
    const T pivot = *f;
    int na = 0, nb = 0;
    for (int i = first + 1; i < last; ++i) if (!lt(v[i], pivot)) alist[na++] = i;
    for (int i = last - 1; i > first; --i) if (!lt(pivot, v[i])) blist[nb++] = i;
    int k = 0;
    while (k < na && k < nb && alist[k] < blist[k]) ++k;
    for (int q = 0; q < k; ++q) sortx::swp(v[alist[q]], v[blist[q]]);
    const int big = 0x7fffffff;
    const int ca = (k < na) ? alist[k] : big, cb = (k > 0) ? blist[k - 1] : big;
    return ca < cb ? ca : cb;
}
template <class T, class C>
RIA_SORT_HD inline void sort_exact_prefix_lists(T* v, int n, int want, int* stack, C lt, int* alist, int* blist, T* tmp) {
    if (n <= 0) return;
    const int limit = (want >= n) ? n : want + 16;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) ++lg;
    int sp = 1;
    stack[0] = 0; stack[1] = n; stack[2] = 2 * lg;
    while (sp > 0) {
        --sp;
        int first = stack[3 * sp], last = stack[3 * sp + 1], depth = stack[3 * sp + 2];
        while (last - first > 16) {
            if (first >= limit) break;
            if (depth == 0) { sortx::heap_sort(v + first, v + last, lt); break; }
            --depth;
            const int cut = partition_lists(v, first, last, lt, alist, blist);
            if (cut < limit) { stack[3 * sp] = cut; stack[3 * sp + 1] = last; stack[3 * sp + 2] = depth; ++sp; }
            last = cut;
        }
    }
    const int fin = (limit < n) ? limit : n;
    for (int i = 0; i < fin; ++i) {          // stable rank sort of the first `fin` positions
        int rank = 0;
        for (int j = 0; j < fin; ++j) rank += (lt(v[j], v[i]) || (j < i && !lt(v[i], v[j]))) ? 1 : 0;
        tmp[rank] = v[i];
    }
    for (int i = 0; i < fin; ++i) v[i] = tmp[i];
}

}  // namespace ria
