// scan.hpp -- device-wide exclusive prefix sum over small per-unit counts.
//
// Used to turn per-block byte counts into stream offsets, per-tile exception
// counts into list offsets, and exception gaps into positions (offset.h:62-67
// is a running sum).  Three launches: per-tile sums, one workgroup scanning
// the tile sums, per-tile rescan.  A tile is 2048 items.
#pragma once

#include "common.hpp"

namespace oroch_hip {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

static inline size_t scan_tiles(size_t n)
{
	return (n + SCAN_TILE - 1) / SCAN_TILE;
}

#ifdef __HIPCC__

// Inclusive scan of one value per thread across a 256-thread workgroup.
// lds: 4 entries.  Returns the inclusive prefix; *all receives the total.
__device__ __forceinline__ uint64_t wg_scan_u64(uint64_t x, uint64_t *lds, uint64_t *all)
{
	const int lane = lane_id(), wave = threadIdx.x / WAVE;
	uint64_t incl = x;
#pragma unroll
	for (int o = 1; o < WAVE; o <<= 1) {
		uint64_t y = __shfl_up(incl, o, WAVE);
		if (lane >= o)
			incl += y;
	}
	if (lane == WAVE - 1)
		lds[wave] = incl;
	__syncthreads();
	uint64_t before = 0, total = 0;
#pragma unroll
	for (int w = 0; w < SCAN_THREADS / WAVE; w++) {
		uint64_t t = lds[w];
		if (w < wave)
			before += t;
		total += t;
	}
	__syncthreads();
	*all = total;
	return incl + before;
}

// ADD is added to every item (gap -> gap + 1 when scanning exception gaps).
template <typename In, int ADD>
__global__ __launch_bounds__(SCAN_THREADS) void k_tile_sums(const In *items, size_t n,
							      uint64_t *tile_sums)
{
	__shared__ uint64_t lds[SCAN_THREADS / WAVE];
	const size_t base = size_t(blockIdx.x) * SCAN_TILE;
	uint64_t s = 0;
	for (int i = 0; i < SCAN_ITEMS; i++) {
		size_t j = base + size_t(i) * SCAN_THREADS + threadIdx.x;
		if (j < n)
			s += uint64_t(items[j]) + ADD;
	}
	uint64_t all;
	wg_scan_u64(s, lds, &all);
	if (threadIdx.x == 0)
		tile_sums[blockIdx.x] = all;
}

// One workgroup: tile_sums[i] <- exclusive prefix; grand total -> *total.
static __global__ __launch_bounds__(SCAN_THREADS) void k_scan_tiles(uint64_t *tile_sums, size_t ntiles,
							       uint64_t *total)
{
	__shared__ uint64_t lds[SCAN_THREADS / WAVE];
	uint64_t carry = 0;
	for (size_t base = 0; base < ntiles; base += SCAN_THREADS) {
		size_t j = base + threadIdx.x;
		uint64_t x = j < ntiles ? tile_sums[j] : 0;
		uint64_t all;
		uint64_t incl = wg_scan_u64(x, lds, &all);
		if (j < ntiles)
			tile_sums[j] = carry + incl - x;
		carry += all;
	}
	if (threadIdx.x == 0)
		*total = carry;
}

// offsets[i] = exclusive prefix of (items + ADD); offsets[n] = total.
template <typename In, int ADD>
__global__ __launch_bounds__(SCAN_THREADS) void k_tile_offsets(const In *items, size_t n,
								 const uint64_t *tile_base,
								 const uint64_t *total, uint64_t *offsets)
{
	__shared__ uint64_t lds[SCAN_THREADS / WAVE];
	const size_t base = size_t(blockIdx.x) * SCAN_TILE + size_t(threadIdx.x) * SCAN_ITEMS;
	uint64_t x[SCAN_ITEMS];
	uint64_t s = 0;
	// a thread's eight items and eight offsets in 16-byte accesses where the
	// arrays allow (whole tiles, 16-byte aligned)
	const bool whole = base + SCAN_ITEMS <= n;
	bool wide_in = false;
	if constexpr (sizeof(In) == 4 && SCAN_ITEMS == 8)
		wide_in = whole && (reinterpret_cast<uintptr_t>(items) & 15u) == 0;
	if (wide_in) {
		const uint4 *q = reinterpret_cast<const uint4 *>(items + base);
		const uint4 a = q[0], b = q[1];
		const uint32_t d[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
		for (int i = 0; i < SCAN_ITEMS; i++) {
			x[i] = uint64_t(d[i]) + ADD;
			s += x[i];
		}
	} else {
#pragma unroll
		for (int i = 0; i < SCAN_ITEMS; i++) {
			x[i] = base + i < n ? uint64_t(items[base + i]) + ADD : 0u;
			s += x[i];
		}
	}
	uint64_t all;
	uint64_t run = tile_base[blockIdx.x] + wg_scan_u64(s, lds, &all) - s;
	if (whole && (reinterpret_cast<uintptr_t>(offsets) & 15u) == 0) {
		uint64_t o[SCAN_ITEMS];
#pragma unroll
		for (int i = 0; i < SCAN_ITEMS; i++) {
			o[i] = run;
			run += x[i];
		}
		uint4 *to = reinterpret_cast<uint4 *>(offsets + base);
#pragma unroll
		for (int i = 0; i < SCAN_ITEMS; i += 2)
			to[i / 2] = make_uint4(uint32_t(o[i]), uint32_t(o[i] >> 32), uint32_t(o[i + 1]), uint32_t(o[i + 1] >> 32));
	} else {
#pragma unroll
		for (int i = 0; i < SCAN_ITEMS; i++) {
			if (base + i < n)
				offsets[base + i] = run;
			run += x[i];
		}
	}
	if (blockIdx.x == 0 && threadIdx.x == 0)
		offsets[n] = *total;
}

// Exclusive scan of items[0..n) (+ADD each) into offsets[0..n], offsets[n] =
// total.  tile_sums: scan_tiles(n) + 1 entries; total: one entry.
template <typename In, int ADD>
static int launch_scan(const In *items, size_t n, uint64_t *tile_sums, uint64_t *total,
		       uint64_t *offsets, hipStream_t st)
{
	const size_t ntiles = scan_tiles(n);
	hipLaunchKernelGGL((k_tile_sums<In, ADD>), dim3(unsigned(ntiles)), dim3(SCAN_THREADS), 0, st,
			   items, n, tile_sums);
	hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(SCAN_THREADS), 0, st, tile_sums, ntiles, total);
	hipLaunchKernelGGL((k_tile_offsets<In, ADD>), dim3(unsigned(ntiles)), dim3(SCAN_THREADS), 0, st,
			   items, n, tile_sums, total, offsets);
	return int(hipGetLastError());
}

#endif // __HIPCC__

} // namespace oroch_hip
