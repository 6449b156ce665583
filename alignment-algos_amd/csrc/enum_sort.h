// enum_sort.h — libstdc++'s std::sort / std::partial_sort restated for ONE lane on a pair of parallel arrays (score, index).
// The pruned enumerators (kscw.h:232-241, crcw.h:314-320) sort their operations with those calls and "higher score first" as
// the only order, so the position of equal scores after the (unstable) sort is observable: the device has to run the very
// same algorithm — introsort + final insertion sort / heap select + sort_heap, GCC 11 bits/stl_algo.h and stl_heap.h.
#pragma once

namespace aln {

namespace kssort {
// element i = (sc[i], ix[i]); comp(a, b) = a.score > b.score   (kscw.h:44-45)
struct Arr {
  float* sc; int* ix;
  __device__ __forceinline__ bool lt(int a, int b) const { return sc[a] > sc[b]; }
  __device__ __forceinline__ void swap(int a, int b) { float s = sc[a]; sc[a] = sc[b]; sc[b] = s; int i = ix[a]; ix[a] = ix[b]; ix[b] = i; }
  __device__ __forceinline__ void move(int dst, int src) { sc[dst] = sc[src]; ix[dst] = ix[src]; }
};
// stl_heap.h __push_heap / __adjust_heap (value passed separately)
__device__ inline void adjust_heap(Arr a, int first, int hole, int len, float vs, int vi) {
  const int top = hole;
  int second = hole;
  while (second < (len - 1) / 2) {
    second = 2 * (second + 1);
    if (a.lt(first + second, first + (second - 1))) second--;
    a.move(first + hole, first + second);
    hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) {
    second = 2 * (second + 1);
    a.move(first + hole, first + (second - 1));
    hole = second - 1;
  }
  int parent = (hole - 1) / 2;                                   // __push_heap
  while (hole > top && a.sc[first + parent] > vs) {
    a.move(first + hole, first + parent);
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a.sc[first + hole] = vs; a.ix[first + hole] = vi;
}
__device__ inline void make_heap(Arr a, int first, int last) {
  const int len = last - first;
  if (len < 2) return;
  int parent = (len - 2) / 2;
  while (true) {
    const float vs = a.sc[first + parent]; const int vi = a.ix[first + parent];
    adjust_heap(a, first, parent, len, vs, vi);
    if (parent == 0) return;
    parent--;
  }
}
__device__ inline void pop_heap(Arr a, int first, int last, int result) {
  const float vs = a.sc[result]; const int vi = a.ix[result];
  a.move(result, first);
  adjust_heap(a, first, 0, last - first, vs, vi);
}
__device__ inline void heap_select(Arr a, int first, int middle, int last) {
  make_heap(a, first, middle);
  for (int i = middle; i < last; ++i)
    if (a.lt(i, first)) pop_heap(a, first, middle, i);
}
__device__ inline void sort_heap(Arr a, int first, int last) {
  while (last - first > 1) { --last; pop_heap(a, first, last, last); }
}
__device__ inline void partial_sort(Arr a, int first, int middle, int last) {   // std::partial_sort
  heap_select(a, first, middle, last);
  sort_heap(a, first, middle);
}
__device__ inline void unguarded_linear_insert(Arr a, int last) {
  const float vs = a.sc[last]; const int vi = a.ix[last];
  int next = last - 1;
  while (vs > a.sc[next]) { a.move(last, next); last = next; --next; }
  a.sc[last] = vs; a.ix[last] = vi;
}
__device__ inline void insertion_sort(Arr a, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (a.lt(i, first)) {
      const float vs = a.sc[i]; const int vi = a.ix[i];
      for (int k = i; k > first; --k) a.move(k, k - 1);          // move_backward(first, i, i + 1)
      a.sc[first] = vs; a.ix[first] = vi;
    } else unguarded_linear_insert(a, i);
  }
}
__device__ inline void move_median_to_first(Arr a, int result, int x, int y, int z) {
  if (a.lt(x, y)) {
    if (a.lt(y, z)) a.swap(result, y);
    else if (a.lt(x, z)) a.swap(result, z);
    else a.swap(result, x);
  } else if (a.lt(x, z)) a.swap(result, x);
  else if (a.lt(y, z)) a.swap(result, z);
  else a.swap(result, y);
}
__device__ inline int unguarded_partition(Arr a, int first, int last, int pivot) {
  while (true) {
    while (a.lt(first, pivot)) ++first;
    --last;
    while (a.lt(pivot, last)) --last;
    if (!(first < last)) return first;
    a.swap(first, last);
    ++first;
  }
}
// std::sort: __introsort_loop (ranges on an explicit stack; the two halves are independent) + __final_insertion_sort
__device__ inline void sort(Arr a, int first, int last) {
  if (first == last) return;
  int n = last - first, lg = 0;
  while ((n >> (lg + 1)) > 0) ++lg;                              // std::__lg
  int stf[64], stl[64], std_[64], sp = 0;
  stf[0] = first; stl[0] = last; std_[0] = 2 * lg; sp = 1;
  while (sp > 0) {
    --sp;
    int f = stf[sp], l = stl[sp], depth = std_[sp];
    while (l - f > 16) {
      if (depth == 0) { partial_sort(a, f, l, l); break; }
      --depth;
      const int mid = f + (l - f) / 2;
      move_median_to_first(a, f, f + 1, mid, l - 1);
      const int cut = unguarded_partition(a, f + 1, l, f);
      if (sp < 64) { stf[sp] = cut; stl[sp] = l; std_[sp] = depth; ++sp; }
      l = cut;
    }
  }
  if (last - first > 16) {
    insertion_sort(a, first, first + 16);
    for (int i = first + 16; i != last; ++i) unguarded_linear_insert(a, i);
  } else insertion_sort(a, first, last);
}
}  // namespace kssort

}  // namespace aln
