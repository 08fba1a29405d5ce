// rulebook.hip — kernels K1/K2: rulebook (indice-pair) construction for
// SubMConv3d(k=3,pad=1), SparseConv3d(k=2,s=2) and its SparseInverseConv3d partner.
// Replaces the indice-pair build inside spconv (network/backbone.py:19-36,74-90,149-152).
//
// Design (MI355X): coordinates -> row lookups go through an open-addressing hash table in HBM/L2
// (64-bit linear keys, linear probing, one atomicCAS per insert); every other step is a coalesced
// streaming pass over a tap-major [K][n_dst] table: lookup -> per-tile ballot counts -> a small scan of
// the workgroup sums -> ballot-rank compaction into pair lists ordered by (tap, dst) (round 3; a
// rocPRIM exclusive scan over the whole table before).  The counts double as the per-tile offset table
// the weight-gradient kernel walks, so no sort is ever needed for SubM rulebooks.  Stride-2 levels:
// occupancy bitmap + popcount prefix (no sort either); the tile order of the large levels: masks by a
// whole-chip launch, each 16384-row block sorted in one workgroup's LDS (rocPRIM block radix sort).
#include "gpn_common.h"  // first: pulls <cstring> ahead of the HIP/rocPRIM headers

#include <rocprim/rocprim.hpp>

namespace {

constexpr int kThreads = 256;
constexpr uint64_t kEmpty = ~0ull;

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

__device__ __forceinline__ uint64_t lin_key(int b, int x, int y, int z, int s0, int s1, int s2) {
  return (((uint64_t)b * (uint64_t)s0 + (uint64_t)x) * (uint64_t)s1 + (uint64_t)y) * (uint64_t)s2 + (uint64_t)z;
}

// (every builder kernel below takes the row count either as an argument or - n_dev != nullptr, gpn::DevRows - from a device
// counter, and walks its elements with a grid stride: one round when the grid was sized from the count)
__global__ void hash_insert_kernel(const int32_t* __restrict__ indices, int64_t N, int s0, int s1, int s2,
                                   uint64_t* __restrict__ hkeys, int32_t* __restrict__ hvals,
                                   uint64_t mask, const int64_t* __restrict__ n_dev) {
  N = gpn::live_rows(n_dev, N);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(indices)[i];
    const uint64_t key = lin_key(c.x, c.y, c.z, c.w, s0, s1, s2);
    uint64_t slot = mix64(key) & mask;
    while (true) {
      unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(hkeys + slot),
                                          (unsigned long long)kEmpty, (unsigned long long)key);
      if (prev == kEmpty || prev == key) {
        // duplicates: the highest row id wins deterministically
        atomicMax(hvals + slot, (int32_t)i);
        break;
      }
      slot = (slot + 1) & mask;
    }
  }
}

__device__ __forceinline__ int32_t hash_find(const uint64_t* __restrict__ hkeys,
                                             const int32_t* __restrict__ hvals, uint64_t mask,
                                             uint64_t key) {
  uint64_t slot = mix64(key) & mask;
  while (true) {
    uint64_t k = hkeys[slot];
    if (k == key) return hvals[slot];
    if (k == kEmpty) return -1;
    slot = (slot + 1) & mask;
  }
}

// table[k][o] = row of the voxel at coord(o)+delta_k, or -1.  One thread per (k,o); o fastest.
__global__ void subm3_lookup_kernel(const int32_t* __restrict__ indices, int64_t N, int s0, int s1, int s2,
                                    const uint64_t* __restrict__ hkeys, const int32_t* __restrict__ hvals,
                                    uint64_t mask, int32_t* __restrict__ table, const int64_t* __restrict__ n_dev) {
  N = gpn::live_rows(n_dev, N);
  if (blockIdx.x == 0 && threadIdx.x == 0) table[27 * N] = -1;  // sentinel entry behind the table
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < 27 * N; t += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(t / N);
    const int64_t o = t - (int64_t)k * N;
    const int4 c = reinterpret_cast<const int4*>(indices)[o];
    const int x = c.y + (k / 9 - 1), y = c.z + ((k / 3) % 3 - 1), z = c.w + (k % 3 - 1);
    int32_t r = -1;
    if (k == 13) {
      r = (int32_t)o;  // centre tap: the row itself
    } else if (x >= 0 && x < s0 && y >= 0 && y < s1 && z >= 0 && z < s2) {
      r = hash_find(hkeys, hvals, mask, lin_key(c.x, x, y, z, s0, s1, s2));
    }
    table[t] = r;
  }
}

struct ValidFlag {
  __device__ __host__ int32_t operator()(int32_t v) const { return v >= 0 ? 1 : 0; }
};

// ---- pair lists from the tap-major table: wave ballots, no device-wide scan over the table (round 3) ------------------------
// pair_src / pair_dst = the valid entries of table [K, n_dst] in (tap, row) order; tile_off[k][t] = position of the first pair
// of tap k at or behind row 32 t; tile_off[k][n_tiles] = end of tap k.  Until round 3: a rocPRIM exclusive scan over all
// K n_dst + 1 entries (4 M at level 0: a 16 MB position array written and read back) + a compaction launch.  Now:
//   A  lists_count_kernel    a half-wave per (tap, 32-row tile): ballot + popcount = the tile's pair count; a workgroup's eight
//                            counts and their sum go to a scratch array
//   B  lists_block_scan_kernel   ONE workgroup: exclusive scan of the workgroup sums (15 k of them at level 0), total
//   C  lists_compact_kernel  same ballots again: position = workgroup offset + prefix of the tile counts before it + rank of
//                            the lane among the tile's valid lanes (ballot mask below the lane) -> pairs in (tap, row) order
// The table is read twice (coalesced); nothing else of its size is written.  Flat tile index f = k * n_tiles + t.
constexpr int kListTilesPerWg = 8;  // 256 threads = 4 waves x 2 half-waves

__device__ __forceinline__ bool lists_entry(const int32_t* __restrict__ table, int K, int64_t n_dst, int64_t n_tiles, int64_t f,
                                            int lane32, int& k, int64_t& row, int32_t& value) {
  k = 0, row = 0, value = -1;
  if (f >= (int64_t)K * n_tiles) return false;
  k = (int)(f / n_tiles);
  const int64_t t = f - (int64_t)k * n_tiles;
  row = t * GPN_TILE_ROWS + lane32;
  if (row >= n_dst) return false;
  value = table[(int64_t)k * n_dst + row];
  return value >= 0;
}

__global__ __launch_bounds__(256) void lists_count_kernel(const int32_t* __restrict__ table, int K, int64_t n_dst, int64_t n_tiles,
                                                          int32_t* __restrict__ counts, int32_t* __restrict__ wg_sum,
                                                          const int64_t* __restrict__ n_dev) {
  __shared__ int32_t c[kListTilesPerWg];
  if (n_dev) {
    n_dst = gpn::live_rows(n_dev, n_dst);
    n_tiles = (n_dst + GPN_TILE_ROWS - 1) / GPN_TILE_ROWS;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int64_t n_wg = ((int64_t)K * n_tiles + kListTilesPerWg - 1) / kListTilesPerWg;
  for (int64_t b = blockIdx.x; b < n_wg; b += gridDim.x) {
    const int64_t f = b * kListTilesPerWg + wave * 2 + half;
    int k;
    int64_t row;
    int32_t v;
    const bool valid = lists_entry(table, K, n_dst, n_tiles, f, lane & 31, k, row, v);
    const uint64_t m = __builtin_amdgcn_ballot_w64(valid);
    const int cnt = __popcll(half ? (m >> 32) : (m & 0xffffffffull));
    if ((lane & 31) == 0) {
      c[wave * 2 + half] = cnt;
      if (f < (int64_t)K * n_tiles) counts[f] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int32_t sum = 0;
#pragma unroll
      for (int i = 0; i < kListTilesPerWg; ++i) sum += c[i];
      wg_sum[b] = sum;
    }
    __syncthreads();  // (c[] is rewritten by the next round)
  }
}

// in place: wg_sum[b] -> exclusive prefix; the grand total to *total (int32) and, if given, *num_pairs (int64)
// (n_dev != nullptr: n and `total` = the end marker of the last tap are derived from the live row count - `total` then points at
// the offset table's base)
__global__ __launch_bounds__(1024) void lists_block_scan_kernel(int32_t* __restrict__ wg_sum, int64_t n, int32_t* __restrict__ total,
                                                                int64_t* __restrict__ num_pairs, int K, int64_t n_dst_bound,
                                                                const int64_t* __restrict__ n_dev) {
  __shared__ int32_t part[1024];
  if (n_dev) {
    const int64_t n_tiles = (gpn::live_rows(n_dev, n_dst_bound) + GPN_TILE_ROWS - 1) / GPN_TILE_ROWS;
    n = ((int64_t)K * n_tiles + kListTilesPerWg - 1) / kListTilesPerWg;
    total += (int64_t)(K - 1) * (n_tiles + 1) + n_tiles;
  }
  const int tid = threadIdx.x;
  const int64_t chunk = (n + 1023) / 1024;
  const int64_t b = tid * chunk, e = b + chunk < n ? b + chunk : n;
  int32_t sum = 0;
  for (int64_t i = b; i < e; ++i) sum += wg_sum[i];
  part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // inclusive scan of the 1024 chunk sums
    const int32_t add = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += add;
    __syncthreads();
  }
  int32_t run = tid > 0 ? part[tid - 1] : 0;
  for (int64_t i = b; i < e; ++i) {
    const int32_t v = wg_sum[i];
    wg_sum[i] = run;
    run += v;
  }
  if (tid == 1023) {
    *total = part[1023];
    if (num_pairs) *num_pairs = part[1023];
  }
}

__global__ __launch_bounds__(256) void lists_compact_kernel(const int32_t* __restrict__ table, int K, int64_t n_dst, int64_t n_tiles,
                                                            const int32_t* __restrict__ counts, const int32_t* __restrict__ wg_off,
                                                            int32_t* __restrict__ pair_src, int32_t* __restrict__ pair_dst,
                                                            int32_t* __restrict__ tile_off, const int64_t* __restrict__ n_dev) {
  if (n_dev) {
    n_dst = gpn::live_rows(n_dev, n_dst);
    n_tiles = (n_dst + GPN_TILE_ROWS - 1) / GPN_TILE_ROWS;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int slot = wave * 2 + half;
  const int64_t F = (int64_t)K * n_tiles;
  const int64_t n_wg = (F + kListTilesPerWg - 1) / kListTilesPerWg;
  for (int64_t b = blockIdx.x; b < n_wg; b += gridDim.x) {
    const int64_t f0 = b * kListTilesPerWg;
    const int64_t f = f0 + slot;
    int k;
    int64_t row;
    int32_t v;
    const bool valid = lists_entry(table, K, n_dst, n_tiles, f, lane & 31, k, row, v);
    const uint64_t m = __builtin_amdgcn_ballot_w64(valid);
    const uint32_t mh = (uint32_t)(half ? (m >> 32) : (m & 0xffffffffull));
    int32_t base = wg_off[b];
    for (int i = 0; i < slot; ++i) base += (f0 + i < F) ? counts[f0 + i] : 0;  // (<= 7 L2-resident reads, the same in every lane)
    if (f < F) {
      if ((lane & 31) == 0) {
        const int64_t t = f - (int64_t)k * n_tiles;
        tile_off[(int64_t)k * (n_tiles + 1) + t] = base;
        if (t == 0 && k > 0) tile_off[(int64_t)(k - 1) * (n_tiles + 1) + n_tiles] = base;  // end of the previous tap
      }
      if (valid) {
        const int32_t p = base + __popc(mh & ((1u << (lane & 31)) - 1u));
        pair_src[p] = v;
        pair_dst[p] = (int32_t)row;
      }
    }
  }
}

__global__ void fill_i32_kernel(int32_t* p, int64_t n, int32_t v) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) p[t] = v;
}
// a[0 .. na) = b[0 .. nb) = v; with device counters: na = ka * *na_dev + 1, nb = kb * *nb_dev + 1 (tables of live size)
__global__ void fill_two_i32_kernel(int32_t* a, int64_t na, int32_t* b, int64_t nb, int32_t v, int ka, const int64_t* na_dev,
                                    int kb, const int64_t* nb_dev) {
  if (na_dev) na = (int64_t)ka * gpn::live_rows(na_dev, (na - 1) / ka) + 1;
  if (nb_dev) nb = (int64_t)kb * gpn::live_rows(nb_dev, (nb - 1) / kb) + 1;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < na + nb; t += (int64_t)gridDim.x * blockDim.x) {
    if (t < na) a[t] = v;
    else b[t - na] = v;
  }
}

size_t scan_temp_bytes(int64_t n) {
  size_t bytes = 0;
  auto in = rocprim::make_transform_iterator((const int32_t*)nullptr, ValidFlag());
  rocprim::exclusive_scan(nullptr, bytes, in, (int32_t*)nullptr, 0, (size_t)(n > 0 ? n : 1),
                          rocprim::plus<int32_t>(), (hipStream_t) nullptr);
  return bytes;
}

// table: [K*n_dst + 1] with table[K*n_dst] == -1.  pos: [K*n_dst + 1].
// scratch (`pos`): K n_tiles tile counts + one sum per workgroup of 8 tiles; callers size it K n_dst + 64 ints
int lists_from_table(const int32_t* table, int32_t* pos, int K, int64_t n_dst, int32_t* pair_src,
                     int32_t* pair_dst, int32_t* tile_off, int64_t* num_pairs, void* prim_tmp,
                     size_t prim_bytes, hipStream_t stream, const gpn::DevRows& rows = gpn::DevRows()) {
  (void)prim_tmp;
  (void)prim_bytes;
  const int64_t n_tiles = gpn::cdiv(n_dst, GPN_TILE_ROWS);
  const int64_t F = (int64_t)K * n_tiles;
  const int64_t n_wg = gpn::cdiv(F, kListTilesPerWg);
  const int64_t plan_wg = gpn::cdiv((int64_t)K * gpn::cdiv(gpn::plan_rows(n_dst, rows), GPN_TILE_ROWS), kListTilesPerWg);
  const unsigned grid = gpn::dev_grid(n_wg, plan_wg, rows.dev != nullptr);
  int32_t* counts = pos;
  int32_t* wg_sum = pos + F;  // (behind the counts of the BOUND: valid for every live count)
  hipLaunchKernelGGL(lists_count_kernel, dim3(grid), dim3(256), 0, stream, table, K, n_dst, n_tiles, counts, wg_sum, rows.dev);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(lists_block_scan_kernel, dim3(1), dim3(1024), 0, stream, wg_sum, n_wg,
                     rows.dev ? tile_off : tile_off + (int64_t)(K - 1) * (n_tiles + 1) + n_tiles, num_pairs, K, n_dst, rows.dev);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(lists_compact_kernel, dim3(grid), dim3(256), 0, stream, table, K, n_dst, n_tiles,
                     (const int32_t*)counts, (const int32_t*)wg_sum, pair_src, pair_dst, tile_off, rows.dev);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

uint64_t hash_capacity(int64_t N) {
  uint64_t cap = 1024;
  while (cap < (uint64_t)(2 * N)) cap <<= 1;
  return cap;
}

// ------------------------------------------------------------------------------------------ down
__global__ void down_keys_kernel(const int32_t* __restrict__ indices, int64_t N, int nb, int o0, int o1, int o2,
                                 uint64_t invalid_key, uint64_t* __restrict__ keys,
                                 uint32_t* __restrict__ vals, int32_t* __restrict__ tap) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int4 c = reinterpret_cast<const int4*>(indices)[i];
  const int x = c.y >> 1, y = c.z >> 1, z = c.w >> 1;
  tap[i] = (c.y & 1) * 4 + (c.z & 1) * 2 + (c.w & 1);
  const bool ok = c.x >= 0 && c.x < nb && c.y >= 0 && c.z >= 0 && c.w >= 0 && x < o0 && y < o1 && z < o2;
  keys[i] = ok ? lin_key(c.x, x, y, z, o0, o1, o2) : invalid_key;
  vals[i] = (uint32_t)i;
}

__global__ void down_flags_kernel(const uint64_t* __restrict__ ks, int64_t N, uint64_t invalid_key,
                                  int32_t* __restrict__ flags) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  uint64_t k = ks[j];
  flags[j] = (k != invalid_key && (j == 0 || ks[j - 1] != k)) ? 1 : 0;
}

__global__ void down_emit_kernel(const uint64_t* __restrict__ ks, const uint32_t* __restrict__ order,
                                 const int32_t* __restrict__ incl, const int32_t* __restrict__ indices,
                                 int64_t N, uint64_t invalid_key, int32_t* __restrict__ out_indices,
                                 int32_t* __restrict__ fine_to_coarse, int64_t* __restrict__ num_out) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  if (j == N - 1) num_out[0] = incl[j];
  const uint64_t k = ks[j];
  const uint32_t i = order[j];
  if (k == invalid_key) {
    fine_to_coarse[i] = -1;
    return;
  }
  const int32_t v = incl[j] - 1;
  fine_to_coarse[i] = v;
  if (j == 0 || ks[j - 1] != k) {
    const int4 c = reinterpret_cast<const int4*>(indices)[i];
    reinterpret_cast<int4*>(out_indices)[v] = make_int4(c.x, c.y >> 1, c.z >> 1, c.w >> 1);
  }
}

__global__ void down_scatter_tables_kernel(const int32_t* __restrict__ fine_to_coarse,
                                           const int32_t* __restrict__ tap, int64_t N, int64_t n_out,
                                           int32_t* __restrict__ tf, int32_t* __restrict__ tb,
                                           const int64_t* __restrict__ n_dev, const int64_t* __restrict__ n_out_dev) {
  N = gpn::live_rows(n_dev, N);
  n_out = gpn::live_rows(n_out_dev, n_out);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t o = fine_to_coarse[i];
    if (o < 0) continue;
    const int k = tap[i];
    tf[(int64_t)k * n_out + o] = (int32_t)i;
    tb[(int64_t)k * N + i] = o;
  }
}

// ---- stride-2 level through an occupancy BITMAP of the coarse grid (round 3): no sort ----------------------------------------
// The coarse rows must come out in ascending linear-key order.  Sorting the N coarse keys of the fine rows (radix sort, a
// dozen launches) is one way; the coarse grid of a level is small (8 scenes x 93 x 98 x 97 cells = 7 M bits = 0.9 MB at the
// first level, less below), so here every fine row sets the bit of its coarse cell, a two-kernel prefix sum over the
// popcounts of the 64-bit words ranks the set bits - bit order IS key order - and each fine row reads its coarse row back
// as prefix[word] + popcount(bits below).  Four small launches, deterministic, identical output.
constexpr int kScanBlock = 1024;  // words per block of the word-popcount scan

__global__ void down_mark_kernel(const int32_t* __restrict__ indices, int64_t N, int nb, int o0, int o1, int o2,
                                 unsigned long long* __restrict__ bitmap, int32_t* __restrict__ tap,
                                 const int64_t* __restrict__ n_dev) {
  N = gpn::live_rows(n_dev, N);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(indices)[i];
    const int x = c.y >> 1, y = c.z >> 1, z = c.w >> 1;
    tap[i] = (c.y & 1) * 4 + (c.z & 1) * 2 + (c.w & 1);
    const bool ok = c.x >= 0 && c.x < nb && c.y >= 0 && c.z >= 0 && c.w >= 0 && x < o0 && y < o1 && z < o2;
    if (!ok) continue;
    const uint64_t key = lin_key(c.x, x, y, z, o0, o1, o2);
    atomicOr(bitmap + (key >> 6), 1ull << (key & 63));
  }
}

// words of the coarse grid's bitmap that can be set: all of them, or - device-counted batch entries (the proposals of a step) -
// those of the first *batch_dev entries
__device__ __forceinline__ int64_t live_words(int64_t n_words, int64_t cells_per_entry, const int64_t* batch_dev) {
  if (!batch_dev) return n_words;
  const int64_t w = (gpn::live_rows(batch_dev, (int64_t)1 << 40) * cells_per_entry + 63) >> 6;
  return w < n_words ? w : n_words;
}
__global__ void bitmap_clear_kernel(unsigned long long* __restrict__ bitmap, int64_t n_words, int64_t cells_per_entry,
                                    const int64_t* __restrict__ batch_dev) {
  n_words = live_words(n_words, cells_per_entry, batch_dev);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * blockDim.x) bitmap[i] = 0ull;
}

// exclusive prefix of the word popcounts inside blocks of kScanBlock words; block totals to `block_sum`
__global__ __launch_bounds__(256) void bitmap_scan_blocks_kernel(const unsigned long long* __restrict__ bitmap, int64_t n_words,
                                                                 int32_t* __restrict__ prefix, int32_t* __restrict__ block_sum,
                                                                 int64_t cells_per_entry, const int64_t* __restrict__ batch_dev) {
  __shared__ int32_t wave_tot[4];
  n_words = live_words(n_words, cells_per_entry, batch_dev);
  const int64_t n_blocks = (n_words + kScanBlock - 1) / kScanBlock;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const int64_t base = blk * kScanBlock + t * 4;
    int32_t c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) c[q] = base + q < n_words ? __popcll(bitmap[base + q]) : 0;
    const int32_t mine = c[0] + c[1] + c[2] + c[3];
    int32_t incl = mine;  // inclusive scan over the wave (ballot-free shuffle tree, fixed order)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int32_t v = __shfl_up(incl, off, 64);
      if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int32_t wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
    int32_t run = wbase + incl - mine;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (base + q < n_words) prefix[base + q] = run;
      run += c[q];
    }
    if (t == 255) block_sum[blk] = wbase + incl;
    __syncthreads();  // (wave_tot is rewritten by the next round)
  }
}

// exclusive scan of the block totals by one workgroup (<= 256 K blocks); total -> num_out
__global__ __launch_bounds__(1024) void bitmap_scan_totals_kernel(int32_t* __restrict__ block_sum, int64_t n_blocks,
                                                                  int64_t* __restrict__ num_out, int64_t n_words,
                                                                  int64_t cells_per_entry, const int64_t* __restrict__ batch_dev) {
  __shared__ int32_t part[1024];
  if (batch_dev) n_blocks = (live_words(n_words, cells_per_entry, batch_dev) + kScanBlock - 1) / kScanBlock;
  const int t = threadIdx.x;
  const int64_t per = (n_blocks + 1023) / 1024;
  const int64_t b = t * per, e = b + per < n_blocks ? b + per : n_blocks;
  int32_t s = 0;
  for (int64_t i = b; i < e; ++i) s += block_sum[i];
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    int32_t run = 0;
    for (int i = 0; i < 1024; ++i) {
      const int32_t v = part[i];
      part[i] = run;
      run += v;
    }
    num_out[0] = run;
  }
  __syncthreads();
  int32_t run = part[t];
  for (int64_t i = b; i < e; ++i) {
    const int32_t v = block_sum[i];
    block_sum[i] = run;
    run += v;
  }
}

__global__ void down_rank_kernel(const int32_t* __restrict__ indices, int64_t N, int nb, int o0, int o1, int o2,
                                 const unsigned long long* __restrict__ bitmap, const int32_t* __restrict__ prefix,
                                 const int32_t* __restrict__ block_sum, int32_t* __restrict__ out_indices,
                                 int32_t* __restrict__ fine_to_coarse, const int64_t* __restrict__ n_dev) {
  N = gpn::live_rows(n_dev, N);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(indices)[i];
    const int x = c.y >> 1, y = c.z >> 1, z = c.w >> 1;
    const bool ok = c.x >= 0 && c.x < nb && c.y >= 0 && c.z >= 0 && c.w >= 0 && x < o0 && y < o1 && z < o2;
    if (!ok) {
      fine_to_coarse[i] = -1;
      continue;
    }
    const uint64_t key = lin_key(c.x, x, y, z, o0, o1, o2);
    const uint64_t word = key >> 6;
    const int32_t v = block_sum[word / kScanBlock] + prefix[word] + __popcll(bitmap[word] & ((1ull << (key & 63)) - 1ull));
    fine_to_coarse[i] = v;
    reinterpret_cast<int4*>(out_indices)[v] = make_int4(c.x, x, y, z);  // (every child writes the same row: idempotent)
  }
}

constexpr uint64_t kBitmapMaxWords = (uint64_t)kScanBlock * 256 * 1024;  // one totals workgroup covers 256 K blocks

size_t sort_temp_bytes(int64_t n) {
  size_t bytes = 0;
  rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                            (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)(n > 0 ? n : 1), 0u, 64u,
                            (hipStream_t) nullptr);
  return bytes;
}
size_t iscan_temp_bytes(int64_t n) {
  size_t bytes = 0;
  rocprim::inclusive_scan(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr,
                          (size_t)(n > 0 ? n : 1), rocprim::plus<int32_t>(), (hipStream_t) nullptr);
  return bytes;
}


// ---- row counts of every coarse level in one pass ------------------------------------------------------------------
// The k2 s2 rule applied l times: level-l coordinate = c >> l, dropped (for good) once it falls outside that level's
// shape (shape_l = shape_{l-1} / 2).  One hash set per level; a voxel whose level-l key is already present stops - the
// voxel that inserted it also inserts the coarser ones.  New keys are counted per wave (ballot) with one atomic each.
__global__ __launch_bounds__(kThreads) void level_counts_kernel(const int32_t* __restrict__ indices, int64_t n_max,
                                                                const int64_t* __restrict__ n_dev, int nb, int s0, int s1,
                                                                int s2, const int64_t* __restrict__ max_coord_dev, int n_levels,
                                                                uint64_t* __restrict__ tables, uint64_t cap,
                                                                int64_t* __restrict__ counts) {
  __shared__ int wg_new[16];  // new keys of this workgroup per level (n_levels <= 16): one global atomic per workgroup and level
  if (threadIdx.x < 16) wg_new[threadIdx.x] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (max_coord_dev) {  // the level-0 extent from the voxeliser's device statistics: max(largest cell index + 1, 128) per axis
    s0 = (int)(max_coord_dev[0] + 1 > 128 ? max_coord_dev[0] + 1 : 128);
    s1 = (int)(max_coord_dev[1] + 1 > 128 ? max_coord_dev[1] + 1 : 128);
    s2 = (int)(max_coord_dev[2] + 1 > 128 ? max_coord_dev[2] + 1 : 128);
  }
  const int64_t n = n_dev ? (*n_dev < n_max ? *n_dev : n_max) : n_max;
  bool active = i < n;
  int4 c = make_int4(0, 0, 0, 0);
  if (active) c = reinterpret_cast<const int4*>(indices)[i];
  active = active && c.x >= 0 && c.x < nb && c.y >= 0 && c.z >= 0 && c.w >= 0 && c.y < s0 && c.z < s1 && c.w < s2;
  int x = c.y, y = c.z, z = c.w;
  const uint64_t mask = cap - 1;
  for (int l = 0; l < n_levels; ++l) {  // wave-uniform trip count: every lane reaches the ballot
    x >>= 1, y >>= 1, z >>= 1;
    s0 /= 2, s1 /= 2, s2 /= 2;
    active = active && x < s0 && y < s1 && z < s2;
    bool is_new = false;
    if (active) {
      const uint64_t key = lin_key(c.x, x, y, z, s0, s1, s2);
      uint64_t* table = tables + (uint64_t)l * cap;
      uint64_t slot = mix64(key) & mask;
      while (true) {
        const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(table + slot),
                                                  (unsigned long long)kEmpty, (unsigned long long)key);
        if (prev == kEmpty) { is_new = true; break; }
        if (prev == key) break;
        slot = (slot + 1) & mask;
      }
    }
    const uint64_t m = __ballot(is_new);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&wg_new[l], (int)__popcll(m));
    active = active && is_new;
  }
  // (an atomic per wave and level on six addresses was half of this kernel's 90 us)
  __syncthreads();
  if (threadIdx.x < n_levels && wg_new[threadIdx.x])
    atomicAdd(reinterpret_cast<unsigned long long*>(counts + threadIdx.x), (unsigned long long)wg_new[threadIdx.x]);
}


// K = 1 "rulebook": every row is its own (only) neighbour.  rows [n], tile_off [n_tiles + 1], nbr [n + 1] (last = -1)
__global__ __launch_bounds__(kThreads) void identity_rulebook_kernel(int64_t n, int64_t n_tiles, int32_t* __restrict__ rows,
                                                                     int32_t* __restrict__ tile_off, int32_t* __restrict__ nbr,
                                                                     int64_t* __restrict__ num_pairs, const int64_t* __restrict__ n_dev) {
  if (n_dev) {
    n = gpn::live_rows(n_dev, n);
    n_tiles = (n + GPN_TILE_ROWS - 1) / GPN_TILE_ROWS;
  }
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i <= n; i += (int64_t)gridDim.x * kThreads) {
    if (i < n) rows[i] = (int32_t)i, nbr[i] = (int32_t)i;
    if (i == n) nbr[n] = -1;
    if (i <= n_tiles) tile_off[i] = (int32_t)(i * GPN_TILE_ROWS < n ? i * GPN_TILE_ROWS : n);
    if (i == 0) num_pairs[0] = n;
  }
}

}  // namespace

// ================================================================================================ subm3
extern "C" size_t gpn_rulebook_subm3_ws_bytes(int64_t N) {
  gpn::WsCarver w(nullptr, 0);
  size_t n = (size_t)(N > 0 ? N : 1);
  w.take<uint64_t>(hash_capacity(N));
  w.take<int32_t>(hash_capacity(N));
  w.take<int32_t>(27 * n + 1);
  w.take<int32_t>(27 * n + 64);
  w.take<char>(scan_temp_bytes(27 * (int64_t)n + 1));
  return w.used;
}

static int rulebook_subm3_impl(const int32_t* indices, int64_t N, const gpn::DevRows& rows, const int32_t* spatial_shape_host,
                               int32_t* nbr, int32_t* pair_src, int32_t* pair_dst, int32_t* tile_off, int64_t* num_pairs, void* ws,
                               size_t ws_bytes, hipStream_t stream);

extern "C" int gpn_rulebook_subm3(const int32_t* indices, int64_t N, const int32_t* spatial_shape_host, int32_t* nbr,
                                  int32_t* pair_src, int32_t* pair_dst, int32_t* tile_off,
                                  int64_t* num_pairs, void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  return rulebook_subm3_impl(indices, N, gpn::DevRows(), spatial_shape_host, nbr, pair_src, pair_dst, tile_off, num_pairs, ws,
                             ws_bytes, (hipStream_t)stream_);
}

// the same with the row count on the device: N = the bound every buffer is sized for (workspace from
// gpn_rulebook_subm3_ws_bytes(N)), *n_dev the live count, n_plan the host's estimate of it (grid sizes only; <= 0: none).
// Outputs are laid out for the LIVE count (nbr [27][*n_dev] + sentinel, tile_off [27][ceil(*n_dev / 32) + 1]).
extern "C" int gpn_rulebook_subm3_dev(const int32_t* indices, int64_t N, const int64_t* n_dev, int64_t n_plan,
                                      const int32_t* spatial_shape_host, int32_t* nbr, int32_t* pair_src, int32_t* pair_dst,
                                      int32_t* tile_off, int64_t* num_pairs, void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  GPN_CHECK_ARG(n_dev != nullptr && N >= 1);
  return rulebook_subm3_impl(indices, N, gpn::DevRows{n_dev, n_plan}, spatial_shape_host, nbr, pair_src, pair_dst, tile_off,
                             num_pairs, ws, ws_bytes, (hipStream_t)stream_);
}

static int rulebook_subm3_impl(const int32_t* indices, int64_t N, const gpn::DevRows& rows, const int32_t* spatial_shape_host,
                               int32_t* nbr, int32_t* pair_src, int32_t* pair_dst, int32_t* tile_off, int64_t* num_pairs, void* ws,
                               size_t ws_bytes, hipStream_t stream) {
  GPN_CHECK_ARG(N >= 0 && spatial_shape_host && tile_off);
  GPN_CHECK_ARG(27 * N + 1 < (int64_t)0x7fffffff);
  const int s0 = spatial_shape_host[0], s1 = spatial_shape_host[1], s2 = spatial_shape_host[2];
  GPN_CHECK_ARG(s0 > 0 && s1 > 0 && s2 > 0);
  const int64_t n_tiles = gpn::cdiv(N, GPN_TILE_ROWS);
  if (N == 0) {
    GPN_CHECK_HIP(hipMemsetAsync(tile_off, 0, sizeof(int32_t) * 27 * (n_tiles + 1), stream));
    if (num_pairs) GPN_CHECK_HIP(hipMemsetAsync(num_pairs, 0, sizeof(int64_t), stream));
    return GPN_OK;
  }
  GPN_CHECK_ARG(indices && pair_src && pair_dst);
  gpn::WsCarver w(ws, ws_bytes);
  const uint64_t cap = hash_capacity(N);
  uint64_t* hkeys = w.take<uint64_t>(cap);
  int32_t* hvals = w.take<int32_t>(cap);
  int32_t* table_ws = w.take<int32_t>(27 * (size_t)N + 1);
  int32_t* table = nbr ? nbr : table_ws;  // the tap-major neighbour table is an output when the caller wants it
  int32_t* pos = w.take<int32_t>(27 * (size_t)N + 64);
  size_t prim_bytes = scan_temp_bytes(27 * N + 1);
  void* prim_tmp = w.take<char>(prim_bytes);
  GPN_CHECK_WS(w);

  gpn::ProfScope prof(GPN_K_RULEBOOK, stream, 0.0, 16.0 * (double)N + 8.0 * 27.0 * (double)N);
  // empty keys (kEmpty = all ones) and values (-1) in ONE fill: the two arrays are adjacent in the workspace
  static_assert(kEmpty == ~0ull, "the key / value tables are initialised by one byte fill");
  GPN_CHECK_HIP(hipMemsetAsync(hkeys, 0xff, (size_t)(reinterpret_cast<char*>(hvals + cap) - reinterpret_cast<char*>(hkeys)), stream));
  const int64_t Np = gpn::plan_rows(N, rows);
  const bool dev = rows.dev != nullptr;
  hipLaunchKernelGGL(hash_insert_kernel, dim3(gpn::dev_grid(gpn::cdiv(N, kThreads), gpn::cdiv(Np, kThreads), dev)), dim3(kThreads), 0,
                     stream, indices, N, s0, s1, s2, hkeys, hvals, cap - 1, rows.dev);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(subm3_lookup_kernel, dim3(gpn::dev_grid(gpn::cdiv(27 * N, kThreads), gpn::cdiv(27 * Np, kThreads), dev)),
                     dim3(kThreads), 0, stream, indices, N, s0, s1, s2, hkeys, hvals, cap - 1, table,
                     rows.dev);  // (also writes the -1 sentinel table[27 N])
  GPN_CHECK_LAUNCH();
  return lists_from_table(table, pos, 27, N, pair_src, pair_dst, tile_off, num_pairs, prim_tmp, prim_bytes,
                          stream, rows);
}

// ================================================================================================ tile order
// Row order for the fused conv's 16-row tiles.  A tile contracts a tap as soon as ANY of its rows has that neighbour, so
// with rows in voxel order a tile of surface voxels executes 23-26 of the 27 taps although a row has 5-13 neighbours.
// Sorting the rows of every block of `block_rows` consecutive rows by their neighbour mask (which taps exist) makes
// tiles mask-homogeneous while keeping them spatially local: 12.6-18 taps per tile at block_rows = 4096 on the bench
// scenes.  Outputs: perm[j] = source row of tile position j (padded to a multiple of 16 + 16 entries), and the
// neighbour table in that order, nbr_p[k][j] = nbr[k][perm[j]].  Stable sort: equal masks keep ascending row order.
__global__ void tile_order_keys_kernel(const int32_t* __restrict__ nbr, int K, int64_t n, int block_shift,
                                       uint64_t* __restrict__ keys, int32_t* __restrict__ vals) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  uint32_t mask = 0;
  for (int k = 0; k < K; ++k) mask |= (nbr[(int64_t)k * n + j] >= 0 ? 1u : 0u) << k;
  keys[j] = ((uint64_t)(j >> block_shift) << 32) | mask;
  vals[j] = (int32_t)j;
}

__global__ void tile_order_gather_kernel(const int32_t* __restrict__ nbr, int32_t* __restrict__ perm, int K,
                                         int64_t n, int64_t padded, int32_t* __restrict__ nbr_p) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < padded - n) perm[n + t] = (int32_t)(n - 1);  // positions past the last row (a partial tile + one spare tile)
  if (t == 0) nbr_p[(int64_t)K * n] = -1;              // sentinel entry behind the table
  if (t >= (int64_t)K * n) return;
  const int64_t k = t / n, j = t - k * n;
  nbr_p[t] = nbr[k * n + perm[j]];
}

// The order is per block of block_rows rows, so a workgroup can sort a block entirely in LDS: mask of its rows (K strided
// table reads per row), rocPRIM's block radix sort over the K mask bits with the row index as value (LSD radix: stable, so
// equal masks keep ascending row order - the same permutation the device-wide sort of (block, mask) keys gives), perm out.
// Two launches (masks by the whole chip, sort by one workgroup per block) instead of the key kernel + ~14 launches of a
// device-wide sort (4 rulebooks per batch take a tile order).  The masks are NOT computed inside the sort kernel: nine
// workgroups reading the 16 MB table of level 0 took 250 us.
#ifndef GPN_TILE_ORDER_RADIX_BITS
#define GPN_TILE_ORDER_RADIX_BITS 8  // digit width of the block sort: 27 mask bits = 4 passes (rocPRIM's default of 4 bits: 7)
#endif
template <int BS, int IPT>
__global__ __launch_bounds__(BS) void tile_order_block_sort_kernel(const uint32_t* __restrict__ mask, int K, int64_t n,
                                                                   int32_t* __restrict__ perm) {
  using sort_t = rocprim::block_radix_sort<uint32_t, BS, IPT, uint32_t, 1, 1, GPN_TILE_ORDER_RADIX_BITS>;
  extern __shared__ __attribute__((aligned(16))) char tile_order_smem[];
  typename sort_t::storage_type& storage = *reinterpret_cast<typename sort_t::storage_type*>(tile_order_smem);
  const int64_t base = (int64_t)blockIdx.x * (BS * IPT) + (int64_t)threadIdx.x * IPT;
  uint32_t keys[IPT], vals[IPT];
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int64_t j = base + i;
    keys[i] = j < n ? mask[j] : 0xffffffffu;  // past the end: behind every real row of the (last) block
    vals[i] = (uint32_t)(j < n ? j : n - 1);
  }
  sort_t().sort(keys, vals, storage, 0, K < 32 ? K : 32);
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int64_t j = base + i;
    if (j < n) perm[j] = (int32_t)vals[i];
  }
}

// the K-bit neighbour mask of every row (whole-chip launch: the table is 4 K n bytes, read coalesced per tap)
__global__ void tile_order_mask_kernel(const int32_t* __restrict__ nbr, int K, int64_t n, uint32_t* __restrict__ mask) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  uint32_t m = 0;
  for (int k = 0; k < K; ++k) m |= (nbr[(int64_t)k * n + j] >= 0 ? 1u : 0u) << k;
  mask[j] = m;
}

template <int BS, int IPT>
int launch_tile_order_block_sort(const uint32_t* mask, int K, int64_t n, int32_t* perm, hipStream_t stream) {
  using sort_t = rocprim::block_radix_sort<uint32_t, BS, IPT, uint32_t, 1, 1, GPN_TILE_ORDER_RADIX_BITS>;
  hipLaunchKernelGGL((tile_order_block_sort_kernel<BS, IPT>), dim3((unsigned)gpn::cdiv(n, (int64_t)BS * IPT)), dim3(BS),
                     sizeof(typename sort_t::storage_type), stream, mask, K, n, perm);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

extern "C" size_t gpn_rulebook_tile_order_ws_bytes(int64_t n) {
  gpn::WsCarver w(nullptr, 0);
  const size_t m = (size_t)(n > 0 ? n : 1);
  w.take<uint64_t>(m);
  w.take<uint64_t>(m);
  w.take<int32_t>(m);
  w.take<char>(sort_temp_bytes(n));
  return w.used;
}

extern "C" int gpn_rulebook_tile_order(const int32_t* nbr, int K, int64_t n, int block_rows, int32_t* perm,
                                       int32_t* nbr_p, void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(K >= 1 && K <= 32 && n >= 1 && nbr && perm && nbr_p);
  GPN_CHECK_ARG(block_rows >= 16 && (block_rows & (block_rows - 1)) == 0);
  int block_shift = 0;
  while ((1 << block_shift) < block_rows) ++block_shift;
  gpn::WsCarver w(ws, ws_bytes);
  uint64_t* keys = w.take<uint64_t>((size_t)n);
  uint64_t* keys_sorted = w.take<uint64_t>((size_t)n);
  int32_t* vals = w.take<int32_t>((size_t)n);
  size_t prim_bytes = sort_temp_bytes(n);
  void* prim_tmp = w.take<char>(prim_bytes);
  GPN_CHECK_WS(w);
  if (block_rows == 16384 || block_rows == 8192 || block_rows == 4096 || block_rows == 2048 || block_rows == 1024) {  // a block fits one workgroup's LDS: one launch
    uint32_t* mask = reinterpret_cast<uint32_t*>(vals);
    hipLaunchKernelGGL(tile_order_mask_kernel, dim3((int)gpn::cdiv(n, kThreads)), dim3(kThreads), 0, stream, nbr, K, n, mask);
    GPN_CHECK_LAUNCH();
    const int rc = block_rows == 16384 ? launch_tile_order_block_sort<1024, 16>(mask, K, n, perm, stream)
                   : block_rows == 8192 ? launch_tile_order_block_sort<512, 16>(mask, K, n, perm, stream)
                   : block_rows == 4096 ? launch_tile_order_block_sort<256, 16>(mask, K, n, perm, stream)
                   : block_rows == 2048 ? launch_tile_order_block_sort<256, 8>(mask, K, n, perm, stream)
                                        : launch_tile_order_block_sort<128, 8>(mask, K, n, perm, stream);
    if (rc != GPN_OK) return rc;
  } else {
    const int grid = (int)gpn::cdiv(n, kThreads);
    hipLaunchKernelGGL(tile_order_keys_kernel, dim3(grid), dim3(kThreads), 0, stream, nbr, K, n, block_shift, keys, vals);
    GPN_CHECK_LAUNCH();
    // key = (block << 32) | mask: only K mask bits and the block bits carry information (5 radix passes instead of 8)
    int block_bits = 1;
    while (((int64_t)1 << block_bits) <= ((n - 1) >> block_shift)) ++block_bits;
    GPN_CHECK_HIP(rocprim::radix_sort_pairs(prim_tmp, prim_bytes, keys, keys_sorted, vals, perm, (size_t)n, 0,
                                            (unsigned)(32 + block_bits), stream));
  }
  const int64_t padded = gpn::cdiv(n, 16) * 16 + 16;
  hipLaunchKernelGGL(tile_order_gather_kernel, dim3((int)gpn::cdiv((int64_t)K * n, kThreads)), dim3(kThreads), 0, stream,
                     nbr, perm, K, n, padded, nbr_p);  // (also pads perm and writes the table's -1 sentinel)
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

// ================================================================================================ down
extern "C" size_t gpn_rulebook_down_ws_bytes(int64_t N) {
  gpn::WsCarver w(nullptr, 0);
  size_t n = (size_t)(N > 0 ? N : 1);
  w.take<uint64_t>(n);
  w.take<uint64_t>(n);
  w.take<uint32_t>(n);
  w.take<uint32_t>(n);
  w.take<int32_t>(n);
  w.take<int32_t>(n);
  size_t a = sort_temp_bytes(N), b = iscan_temp_bytes(N);
  w.take<char>(a > b ? a : b);
  return w.used;
}

static int rulebook_down_impl(const int32_t* indices, int64_t N, const gpn::DevRows& rows, int64_t batch_size,
                              const gpn::DevRows& batch, const int32_t* spatial_shape_host, int32_t* out_indices,
                              int32_t* fine_to_coarse, int32_t* tap, int64_t* num_out, void* ws, size_t ws_bytes, hipStream_t stream);

extern "C" int gpn_rulebook_down(const int32_t* indices, int64_t N, int64_t batch_size,
                                 const int32_t* spatial_shape_host, int32_t* out_indices, int32_t* fine_to_coarse, int32_t* tap,
                                 int64_t* num_out, void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  return rulebook_down_impl(indices, N, gpn::DevRows(), batch_size, gpn::DevRows(), spatial_shape_host, out_indices, fine_to_coarse,
                            tap, num_out, ws, ws_bytes, (hipStream_t)stream_);
}

// row count and batch-entry count (the proposals of a step) on the device: N / batch_size are the bounds the buffers are sized
// for, *_plan the host's estimates (grid sizes).  Needs the bitmap form (workspace sized for the bounds).
extern "C" int gpn_rulebook_down_dev(const int32_t* indices, int64_t N, const int64_t* n_dev, int64_t n_plan, int64_t batch_size,
                                     const int64_t* batch_dev, int64_t batch_plan, const int32_t* spatial_shape_host,
                                     int32_t* out_indices, int32_t* fine_to_coarse, int32_t* tap, int64_t* num_out, void* ws,
                                     size_t ws_bytes, gpn_stream_t stream_) {
  GPN_CHECK_ARG(n_dev != nullptr && N >= 1);
  return rulebook_down_impl(indices, N, gpn::DevRows{n_dev, n_plan}, batch_size, gpn::DevRows{batch_dev, batch_plan},
                            spatial_shape_host, out_indices, fine_to_coarse, tap, num_out, ws, ws_bytes, (hipStream_t)stream_);
}

extern "C" size_t gpn_rulebook_down_dev_ws_bytes(int64_t N, int64_t batch_size, const int32_t* spatial_shape_host) {
  if (!spatial_shape_host) return 0;
  const uint64_t cells = (uint64_t)(spatial_shape_host[0] / 2) * (uint64_t)(spatial_shape_host[1] / 2) * (uint64_t)(spatial_shape_host[2] / 2);
  const uint64_t n_words = ((uint64_t)(batch_size > 0 ? batch_size : 1) * cells + 63) / 64;
  gpn::WsCarver w(nullptr, 0);
  w.take<unsigned long long>((size_t)n_words);
  w.take<int32_t>((size_t)n_words);
  w.take<int32_t>((size_t)gpn::cdiv((int64_t)n_words, kScanBlock));
  const size_t a = w.used, b = gpn_rulebook_down_ws_bytes(N);
  return a > b ? a : b;
}

static int rulebook_down_impl(const int32_t* indices, int64_t N, const gpn::DevRows& rows, int64_t batch_size,
                              const gpn::DevRows& batch, const int32_t* spatial_shape_host, int32_t* out_indices,
                              int32_t* fine_to_coarse, int32_t* tap, int64_t* num_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  GPN_CHECK_ARG(N >= 0 && spatial_shape_host && num_out);
  if (N == 0) {
    GPN_CHECK_HIP(hipMemsetAsync(num_out, 0, sizeof(int64_t), stream));
    return GPN_OK;
  }
  GPN_CHECK_ARG(indices && out_indices && fine_to_coarse && tap);
  GPN_CHECK_ARG(N < (int64_t)0x7fffffff);
  const int o0 = spatial_shape_host[0] / 2, o1 = spatial_shape_host[1] / 2, o2 = spatial_shape_host[2] / 2;
  GPN_CHECK_ARG(o0 > 0 && o1 > 0 && o2 > 0);
  GPN_CHECK_ARG(batch_size >= 1);
  const uint64_t batch_bound = (uint64_t)batch_size;  // rows with batch >= batch_size are dropped
  long double total = (long double)batch_bound * o0 * o1 * o2;
  GPN_CHECK_ARG(total < 9.0e18L);
  const uint64_t invalid_key = batch_bound * (uint64_t)o0 * (uint64_t)o1 * (uint64_t)o2;
  unsigned key_bits = 1;
  while (key_bits < 64 && (invalid_key >> key_bits) != 0) ++key_bits;

  // the coarse grid as a bitmap when it is small enough (always, for the network's levels): no sort
  const uint64_t n_words = (invalid_key + 63) / 64;
  const int64_t n_blocks = gpn::cdiv((int64_t)n_words, kScanBlock);
  gpn::WsCarver wb(ws, ws_bytes);
  unsigned long long* bitmap = wb.take<unsigned long long>((size_t)n_words);
  int32_t* prefix = wb.take<int32_t>((size_t)n_words);
  int32_t* block_sum = wb.take<int32_t>((size_t)n_blocks);
  if (n_words <= kBitmapMaxWords && wb.ok()) {
    const bool dev = rows.dev != nullptr;
    const unsigned gridN = gpn::dev_grid(gpn::cdiv(N, kThreads), gpn::cdiv(gpn::plan_rows(N, rows), kThreads), dev);
    const int64_t cells = (int64_t)o0 * o1 * o2;
    // (bitmap words that can be set: those of the batch entries that exist - all of them unless their count is on the device)
    const int64_t plan_words = gpn::cdiv(gpn::plan_rows(batch_size, batch) * cells, 64);
    const unsigned grid_blocks = gpn::dev_grid(n_blocks, gpn::cdiv(plan_words, kScanBlock), batch.dev != nullptr, 1, 256);
    gpn::ProfScope prof(GPN_K_RULEBOOK, stream, 0.0, 16.0 * (double)N * 2 + 8.0 * (double)N);
    if (batch.dev) {
      hipLaunchKernelGGL(bitmap_clear_kernel, dim3(gpn::dev_grid(gpn::cdiv((int64_t)n_words, kThreads), gpn::cdiv(plan_words, kThreads), true)),
                         dim3(kThreads), 0, stream, bitmap, (int64_t)n_words, cells, batch.dev);
      GPN_CHECK_LAUNCH();
    } else {
      GPN_CHECK_HIP(hipMemsetAsync(bitmap, 0, (size_t)n_words * sizeof(unsigned long long), stream));
    }
    hipLaunchKernelGGL(down_mark_kernel, dim3(gridN), dim3(kThreads), 0, stream, indices, N, (int)batch_size, o0, o1, o2, bitmap, tap,
                       rows.dev);
    GPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(bitmap_scan_blocks_kernel, dim3(grid_blocks), dim3(256), 0, stream, bitmap, (int64_t)n_words, prefix,
                       block_sum, cells, batch.dev);
    GPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(bitmap_scan_totals_kernel, dim3(1), dim3(1024), 0, stream, block_sum, n_blocks, num_out, (int64_t)n_words,
                       cells, batch.dev);
    GPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(down_rank_kernel, dim3(gridN), dim3(kThreads), 0, stream, indices, N, (int)batch_size, o0, o1, o2, bitmap,
                       prefix, block_sum, out_indices, fine_to_coarse, rows.dev);
    GPN_CHECK_LAUNCH();
    return GPN_OK;
  }
  if (rows.dev || batch.dev) {
    gpn::set_error("gpn_rulebook_down_dev: the coarse grid does not fit the bitmap form (workspace of gpn_rulebook_down_dev_ws_bytes?)");
    return GPN_ERR_WS;
  }
  gpn::WsCarver w(ws, ws_bytes);
  uint64_t* keys = w.take<uint64_t>((size_t)N);
  uint64_t* ks = w.take<uint64_t>((size_t)N);
  uint32_t* vals = w.take<uint32_t>((size_t)N);
  uint32_t* order = w.take<uint32_t>((size_t)N);
  int32_t* flags = w.take<int32_t>((size_t)N);
  int32_t* incl = w.take<int32_t>((size_t)N);
  size_t a = sort_temp_bytes(N), b = iscan_temp_bytes(N);
  size_t prim_bytes = a > b ? a : b;
  void* prim_tmp = w.take<char>(prim_bytes);
  GPN_CHECK_WS(w);

  const int grid = (int)gpn::cdiv(N, kThreads);
  gpn::ProfScope prof(GPN_K_RULEBOOK, stream, 0.0, 16.0 * (double)N * 2 + 8.0 * (double)N);
  hipLaunchKernelGGL(down_keys_kernel, dim3(grid), dim3(kThreads), 0, stream, indices, N, (int)batch_size, o0,
                     o1, o2, invalid_key, keys, vals, tap);
  GPN_CHECK_LAUNCH();
  size_t tmp = prim_bytes;
  GPN_CHECK_HIP(rocprim::radix_sort_pairs(prim_tmp, tmp, keys, ks, vals, order, (size_t)N, 0u, key_bits, stream));
  hipLaunchKernelGGL(down_flags_kernel, dim3(grid), dim3(kThreads), 0, stream, ks, N, invalid_key, flags);
  GPN_CHECK_LAUNCH();
  tmp = prim_bytes;
  GPN_CHECK_HIP(rocprim::inclusive_scan(prim_tmp, tmp, flags, incl, (size_t)N, rocprim::plus<int32_t>(), stream));
  hipLaunchKernelGGL(down_emit_kernel, dim3(grid), dim3(kThreads), 0, stream, ks, order, incl, indices, N,
                     invalid_key, out_indices, fine_to_coarse, num_out);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

extern "C" size_t gpn_rulebook_down_lists_ws_bytes(int64_t N, int64_t n_out) {
  gpn::WsCarver w(nullptr, 0);
  size_t n = (size_t)(N > 0 ? N : 1), no = (size_t)(n_out > 0 ? n_out : 1);
  w.take<int32_t>(8 * no + 1);
  w.take<int32_t>(8 * n + 1);
  w.take<int32_t>(8 * n + 64);  // scratch of the list builder (shared, sized for the larger table)
  w.take<char>(scan_temp_bytes(8 * (int64_t)n + 1));
  return w.used;
}

static int rulebook_down_lists_impl(const int32_t* fine_to_coarse, const int32_t* tap, int64_t N, const gpn::DevRows& rows,
                                    int64_t n_out, const gpn::DevRows& rows_out, int32_t* fwd_nbr, int32_t* fwd_src, int32_t* fwd_dst,
                                    int32_t* fwd_tile_off, int32_t* bwd_nbr, int32_t* bwd_src, int32_t* bwd_dst,
                                    int32_t* bwd_tile_off, int64_t* num_pairs, void* ws, size_t ws_bytes, hipStream_t stream);

extern "C" int gpn_rulebook_down_lists(const int32_t* fine_to_coarse, const int32_t* tap, int64_t N,
                                       int64_t n_out, int32_t* fwd_nbr, int32_t* fwd_src, int32_t* fwd_dst,
                                       int32_t* fwd_tile_off, int32_t* bwd_nbr, int32_t* bwd_src, int32_t* bwd_dst,
                                       int32_t* bwd_tile_off, int64_t* num_pairs, void* ws, size_t ws_bytes,
                                       gpn_stream_t stream_) {
  return rulebook_down_lists_impl(fine_to_coarse, tap, N, gpn::DevRows(), n_out, gpn::DevRows(), fwd_nbr, fwd_src, fwd_dst,
                                  fwd_tile_off, bwd_nbr, bwd_src, bwd_dst, bwd_tile_off, num_pairs, ws, ws_bytes, (hipStream_t)stream_);
}

// fine and coarse row counts on the device (N, n_out: the bounds; tables laid out for the live counts)
extern "C" int gpn_rulebook_down_lists_dev(const int32_t* fine_to_coarse, const int32_t* tap, int64_t N, const int64_t* n_dev,
                                           int64_t n_plan, int64_t n_out, const int64_t* n_out_dev, int64_t n_out_plan,
                                           int32_t* fwd_nbr, int32_t* fwd_src, int32_t* fwd_dst, int32_t* fwd_tile_off,
                                           int32_t* bwd_nbr, int32_t* bwd_src, int32_t* bwd_dst, int32_t* bwd_tile_off,
                                           int64_t* num_pairs, void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  GPN_CHECK_ARG(n_dev && n_out_dev && N >= 1 && n_out >= 1);
  return rulebook_down_lists_impl(fine_to_coarse, tap, N, gpn::DevRows{n_dev, n_plan}, n_out, gpn::DevRows{n_out_dev, n_out_plan},
                                  fwd_nbr, fwd_src, fwd_dst, fwd_tile_off, bwd_nbr, bwd_src, bwd_dst, bwd_tile_off, num_pairs, ws,
                                  ws_bytes, (hipStream_t)stream_);
}

static int rulebook_down_lists_impl(const int32_t* fine_to_coarse, const int32_t* tap, int64_t N, const gpn::DevRows& rows,
                                    int64_t n_out, const gpn::DevRows& rows_out, int32_t* fwd_nbr, int32_t* fwd_src, int32_t* fwd_dst,
                                    int32_t* fwd_tile_off, int32_t* bwd_nbr, int32_t* bwd_src, int32_t* bwd_dst,
                                    int32_t* bwd_tile_off, int64_t* num_pairs, void* ws, size_t ws_bytes, hipStream_t stream) {
  GPN_CHECK_ARG(N >= 0 && n_out >= 0 && n_out <= N && fwd_tile_off && bwd_tile_off);
  const int64_t nt_f = gpn::cdiv(n_out, GPN_TILE_ROWS), nt_b = gpn::cdiv(N, GPN_TILE_ROWS);
  if (N == 0 || n_out == 0) {
    GPN_CHECK_HIP(hipMemsetAsync(fwd_tile_off, 0, sizeof(int32_t) * 8 * (nt_f + 1), stream));
    GPN_CHECK_HIP(hipMemsetAsync(bwd_tile_off, 0, sizeof(int32_t) * 8 * (nt_b + 1), stream));
    if (num_pairs) GPN_CHECK_HIP(hipMemsetAsync(num_pairs, 0, sizeof(int64_t), stream));
    return GPN_OK;
  }
  GPN_CHECK_ARG(fine_to_coarse && tap && fwd_src && fwd_dst && bwd_src && bwd_dst);
  gpn::WsCarver w(ws, ws_bytes);
  int32_t* tf_ws = w.take<int32_t>(8 * (size_t)n_out + 1);
  int32_t* tb_ws = w.take<int32_t>(8 * (size_t)N + 1);
  int32_t* tf = fwd_nbr ? fwd_nbr : tf_ws;
  int32_t* tb = bwd_nbr ? bwd_nbr : tb_ws;
  int32_t* pos = w.take<int32_t>(8 * (size_t)N + 64);
  size_t prim_bytes = scan_temp_bytes(8 * N + 1);
  void* prim_tmp = w.take<char>(prim_bytes);
  GPN_CHECK_WS(w);
  gpn::ProfScope prof(GPN_K_RULEBOOK, stream, 0.0, 8.0 * (double)N + 16.0 * (double)N);
  const bool dev = rows.dev != nullptr;
  const int64_t Np = gpn::plan_rows(N, rows), Nop = gpn::plan_rows(n_out, rows_out);
  {  // both tables to -1 in one launch (they are separate caller-owned outputs: two memsets otherwise)
    const int64_t nf = 8 * n_out + 1, nb = 8 * N + 1;
    hipLaunchKernelGGL(fill_two_i32_kernel, dim3(gpn::dev_grid(gpn::cdiv(nf + nb, kThreads), gpn::cdiv(8 * (Np + Nop) + 2, kThreads), dev)),
                       dim3(kThreads), 0, stream, tf, nf, tb, nb, -1, 8, rows_out.dev, 8, rows.dev);
    GPN_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(down_scatter_tables_kernel, dim3(gpn::dev_grid(gpn::cdiv(N, kThreads), gpn::cdiv(Np, kThreads), dev)), dim3(kThreads), 0,
                     stream, fine_to_coarse, tap, N, n_out, tf, tb, rows.dev, rows_out.dev);
  GPN_CHECK_LAUNCH();
  int rc = lists_from_table(tf, pos, 8, n_out, fwd_src, fwd_dst, fwd_tile_off, num_pairs, prim_tmp, prim_bytes,
                            stream, rows_out);
  if (rc != GPN_OK) return rc;
  return lists_from_table(tb, pos, 8, N, bwd_src, bwd_dst, bwd_tile_off, nullptr, prim_tmp, prim_bytes, stream, rows);
}


static uint64_t level_table_cap(int64_t n_max) {
  uint64_t cap = 1024;
  while (cap < 2 * (uint64_t)(n_max > 0 ? n_max : 1)) cap <<= 1;
  return cap;
}

extern "C" size_t gpn_rulebook_level_counts_ws_bytes(int64_t n_max, int n_levels) {
  return gpn::align_up((size_t)level_table_cap(n_max) * (size_t)(n_levels > 0 ? n_levels : 1) * sizeof(uint64_t));
}

extern "C" int gpn_rulebook_level_counts(const int32_t* indices, int64_t n_max, const int64_t* n_dev, int64_t batch_size,
                                         const int32_t* spatial_shape_host, int n_levels, int64_t* counts, void* ws,
                                         size_t ws_bytes, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(n_max >= 0 && n_levels >= 1 && n_levels <= 16 && spatial_shape_host && counts && batch_size >= 1);
  GPN_CHECK_HIP(hipMemsetAsync(counts, 0, sizeof(int64_t) * (size_t)n_levels, stream));
  if (n_max == 0) return GPN_OK;
  GPN_CHECK_ARG(indices && n_max < (int64_t)0x7fffffff);
  const uint64_t cap = level_table_cap(n_max);
  const size_t need = (size_t)cap * (size_t)n_levels * sizeof(uint64_t);
  if (!ws || ws_bytes < need) {
    gpn::set_error("gpn_rulebook_level_counts: workspace too small (%zu needed, %zu given)", need, ws_bytes);
    return GPN_ERR_WS;
  }
  GPN_CHECK_HIP(hipMemsetAsync(ws, 0xff, need, stream));
  gpn::ProfScope prof(GPN_K_RULEBOOK, stream, 0.0, 16.0 * (double)n_max);
  hipLaunchKernelGGL(level_counts_kernel, dim3((unsigned)gpn::cdiv(n_max, kThreads)), dim3(kThreads), 0, stream, indices, n_max,
                     n_dev, (int)batch_size, spatial_shape_host[0], spatial_shape_host[1], spatial_shape_host[2], nullptr, n_levels,
                     static_cast<uint64_t*>(ws), cap, counts);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

// the same with the row count and the level-0 extent still on the device (gpn_voxelize_scenes: no host read in between);
// counts must be zeroed by the caller
int gpn::rulebook_level_counts_dev(const int32_t* indices, int64_t n_max, const int64_t* n_dev, int64_t batch_size,
                                   const int64_t* max_coord_dev, int n_levels, int64_t* counts, void* ws, size_t ws_bytes,
                                   hipStream_t stream) {
  GPN_CHECK_ARG(indices && n_dev && max_coord_dev && counts && n_levels >= 1 && n_levels <= 16 && n_max >= 1);
  const uint64_t cap = level_table_cap(n_max);
  const size_t need = (size_t)cap * (size_t)n_levels * sizeof(uint64_t);
  if (!ws || ws_bytes < need) {
    gpn::set_error("gpn_voxelize_scenes: workspace too small for the level counts (%zu needed, %zu given)", need, ws_bytes);
    return GPN_ERR_WS;
  }
  GPN_CHECK_HIP(hipMemsetAsync(ws, 0xff, need, stream));
  gpn::ProfScope prof(GPN_K_RULEBOOK, stream, 0.0, 16.0 * (double)n_max);
  hipLaunchKernelGGL(level_counts_kernel, dim3((unsigned)gpn::cdiv(n_max, kThreads)), dim3(kThreads), 0, stream, indices, n_max,
                     n_dev, (int)batch_size, 0, 0, 0, max_coord_dev, n_levels, static_cast<uint64_t*>(ws), cap, counts);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

extern "C" int gpn_rulebook_identity(int64_t n, int32_t* rows, int32_t* tile_off, int32_t* nbr, int64_t* num_pairs,
                                     gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(n >= 0 && n < (int64_t)0x7fffffff && rows && tile_off && nbr && num_pairs);
  const int64_t n_tiles = gpn::cdiv(n, (int64_t)GPN_TILE_ROWS);
  hipLaunchKernelGGL(identity_rulebook_kernel, dim3((unsigned)gpn::cdiv(n + 1, (int64_t)kThreads)), dim3(kThreads), 0, stream, n,
                     n_tiles, rows, tile_off, nbr, num_pairs, (const int64_t*)nullptr);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

// the same for a device-counted row count (n = the bound of the buffers)
extern "C" int gpn_rulebook_identity_dev(int64_t n, const int64_t* n_dev, int64_t n_plan, int32_t* rows, int32_t* tile_off,
                                         int32_t* nbr, int64_t* num_pairs, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(n >= 1 && n < (int64_t)0x7fffffff && n_dev && rows && tile_off && nbr && num_pairs);
  const int64_t np = gpn::plan_rows(n, gpn::DevRows{n_dev, n_plan});
  hipLaunchKernelGGL(identity_rulebook_kernel, dim3(gpn::dev_grid(gpn::cdiv(n + 1, (int64_t)kThreads), gpn::cdiv(np + 1, (int64_t)kThreads), true)),
                     dim3(kThreads), 0, stream, n, gpn::cdiv(n, (int64_t)GPN_TILE_ROWS), rows, tile_off, nbr, num_pairs, n_dev);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}
