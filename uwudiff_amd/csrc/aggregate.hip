// Ragged -> padded aggregation of per-caption embeddings (conditioning front-end).
// Reference: src/duwu/utils/aggregation.py:6-171 -- concat_aggregate_embeddings(_vectorize) scatters the n_b
// consecutive [seq, ...] embeddings of image b into row b of a [B, max_n * seq, ...] tensor filled with pad_value,
// split_aggregate_embeddings is its inverse, first_aggregate_embeddings keeps the first embedding of each image.
// Pure byte movement: for image b the source elements start_b .. start_b + n_b - 1 are contiguous and so is their
// destination, so each output row is one contiguous copy of n_b * unit bytes followed by padding.  One thread moves
// one vector of V bytes (16 when unit and the base pointers allow it, else 4 / 2 / 1); HBM-bound, coalesced.
#include "common.h"

namespace {

template <int V> struct Vec;
template <> struct Vec<16> { typedef uint4 type; };
template <> struct Vec<8> { typedef uint2 type; };
template <> struct Vec<4> { typedef uint32_t type; };
template <> struct Vec<2> { typedef uint16_t type; };
template <> struct Vec<1> { typedef uint8_t type; };

template <int V>
__device__ __forceinline__ typename Vec<V>::type pad_vec(uint64_t pad_bits, int elem_size) {
  // pad pattern: the element's bytes repeated (elem_size 1/2/4/8 divides or is divided by V)
  uint64_t lo = pad_bits;
  if (elem_size == 1) lo = (lo & 0xff) * 0x0101010101010101ull;
  else if (elem_size == 2) lo = (lo & 0xffff) * 0x0001000100010001ull;
  else if (elem_size == 4) lo = (lo & 0xffffffffull) * 0x0000000100000001ull;
  typename Vec<V>::type v;
  if constexpr (V == 16) v = uint4{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)lo, (uint32_t)(lo >> 32)};
  else if constexpr (V == 8) v = uint2{(uint32_t)lo, (uint32_t)(lo >> 32)};
  else v = (typename Vec<V>::type)lo;
  return v;
}

// MODE 0: concat (gather src -> padded dst), 1: split (padded src -> ragged dst), 2: first
template <int V, int MODE>
__global__ void __launch_bounds__(256) aggregate_kernel(const char* __restrict__ src, const int* __restrict__ starts,
                                                        char* __restrict__ dst, int B, int64_t row_vecs,
                                                        int64_t unit_vecs, uint64_t pad_bits, int elem_size) {
  typedef typename Vec<V>::type VT;
  const int64_t total = (int64_t)B * row_vecs;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
    const int b = (int)(g / row_vecs);
    const int64_t o = g - (int64_t)b * row_vecs;  // vector index inside the padded row
    const int s0 = starts[b], n = starts[b + 1] - s0;
    if constexpr (MODE == 0) {
      VT v;
      if (o < (int64_t)n * unit_vecs) v = reinterpret_cast<const VT*>(src)[(int64_t)s0 * unit_vecs + o];
      else v = pad_vec<V>(pad_bits, elem_size);
      reinterpret_cast<VT*>(dst)[g] = v;
    } else if constexpr (MODE == 1) {
      if (o < (int64_t)n * unit_vecs) reinterpret_cast<VT*>(dst)[(int64_t)s0 * unit_vecs + o] = reinterpret_cast<const VT*>(src)[g];
    } else {  // row_vecs == unit_vecs: dst[b] = src[s0]
      reinterpret_cast<VT*>(dst)[g] = reinterpret_cast<const VT*>(src)[(int64_t)s0 * unit_vecs + o];
    }
  }
}

template <int MODE>
int launch_agg(const void* src, const int* starts, void* dst, int B, int64_t row_bytes, int64_t unit_bytes,
               uint64_t pad_bits, int elem_size, hipStream_t st) {
  const uintptr_t al = (uintptr_t)src | (uintptr_t)dst | (uintptr_t)unit_bytes | (uintptr_t)row_bytes;
  int V = 1;
  if ((al & 15) == 0) V = 16;
  else if ((al & 7) == 0) V = 8;
  else if ((al & 3) == 0) V = 4;
  else if ((al & 1) == 0) V = 2;
  if (V < elem_size) V = elem_size;  // (tensors are element-aligned: cannot happen for torch allocations)
  const int64_t total = (int64_t)B * (row_bytes / V);
  const int grid = ew_grid(total, 256);
#define AG(VV) hipLaunchKernelGGL((aggregate_kernel<VV, MODE>), dim3(grid), dim3(256), 0, st, (const char*)src, starts, \
                                  (char*)dst, B, row_bytes / VV, unit_bytes / VV, pad_bits, elem_size)
  switch (V) {
    case 16: AG(16); break;
    case 8: AG(8); break;
    case 4: AG(4); break;
    case 2: AG(2); break;
    default: AG(1); break;
  }
#undef AG
  return UWU_OK;
}

}  // namespace

extern "C" int uwu_aggregate_concat(const void* emb, const int* starts, void* out, int B, int max_n, int64_t unit_bytes,
                                    int elem_size, uint64_t pad_bits, void* stream) {
  UWU_CHECK_ARG(emb && starts && out && B > 0 && max_n > 0 && unit_bytes > 0, "aggregate_concat: bad args");
  UWU_CHECK_ARG(elem_size == 1 || elem_size == 2 || elem_size == 4 || elem_size == 8, "aggregate_concat: elem_size %d",
                elem_size);
  UWU_CHECK_ARG(unit_bytes % elem_size == 0, "aggregate_concat: unit_bytes not a multiple of elem_size");
  const int rc = launch_agg<0>(emb, starts, out, B, (int64_t)max_n * unit_bytes, unit_bytes, pad_bits, elem_size,
                               (hipStream_t)stream);
  UWU_LAUNCH_CHECK("aggregate_concat");
  return rc;
}

extern "C" int uwu_aggregate_split(const void* cat, const int* starts, void* out, int B, int max_n, int64_t unit_bytes,
                                   void* stream) {
  UWU_CHECK_ARG(cat && starts && out && B > 0 && max_n > 0 && unit_bytes > 0, "aggregate_split: bad args");
  const int rc = launch_agg<1>(cat, starts, out, B, (int64_t)max_n * unit_bytes, unit_bytes, 0, 1, (hipStream_t)stream);
  UWU_LAUNCH_CHECK("aggregate_split");
  return rc;
}

extern "C" int uwu_aggregate_first(const void* emb, const int* starts, void* out, int B, int64_t unit_bytes, void* stream) {
  UWU_CHECK_ARG(emb && starts && out && B > 0 && unit_bytes > 0, "aggregate_first: bad args");
  const int rc = launch_agg<2>(emb, starts, out, B, unit_bytes, unit_bytes, 0, 1, (hipStream_t)stream);
  UWU_LAUNCH_CHECK("aggregate_first");
  return rc;
}
