// chain.hpp -- chained scan with decoupled look-back: a tile learns the sum
// of the counts of all tiles before it from one 8-byte word per tile.
//
// state[t]: bits 63..62 = 0 nothing yet, 1 = count of tile t, 2 = counts of
// tiles 0..t; bits 61..0 the count.  One naturally aligned 8-byte word written
// by one write-through store and polled with cache-bypassing loads, so it
// needs no fence.  The array starts out as zeros.
#pragma once

#include "common.hpp"

namespace oroch_hip {

#ifdef __HIPCC__

constexpr uint64_t ST_AGG = uint64_t(1) << 62, ST_PREFIX = uint64_t(2) << 62;
constexpr uint64_t ST_VALUE = (uint64_t(1) << 62) - 1;
constexpr uint32_t SPIN_LIMIT = 1u << 18;

__device__ __forceinline__ void state_store(uint64_t *p, uint64_t v)
{
	__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t state_load(const uint64_t *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sum of the byte counts of tiles 0..tile-1, by one wavefront: lane l looks at
// tile (tile - 1 - l) of the current window of 64; the nearest tile holding a
// running total ends the walk.  Returns false if a wait ran out.
__device__ __forceinline__ bool look_back(const uint64_t *state, size_t tile, int lane,
					  const uint32_t *abort_flag, uint64_t &exclusive)
{
	uint64_t sum = 0;
	long long win = (long long) tile - 1;
	for (;;) {
		const long long idx = win - lane;
		uint64_t w;
		uint32_t spins = 0;
		for (;;) {
			w = idx >= 0 ? state_load(&state[idx]) : ST_PREFIX;
			if (__ballot((w >> 62) == 0) == 0)
				break;
			spins++;
			if (spins > SPIN_LIMIT)
				return false;
			if ((spins & 255u) == 0 &&
			    __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
				return false;
			__builtin_amdgcn_s_sleep(2);
		}
		const uint64_t have_prefix = __ballot((w >> 62) == 2);
		const uint32_t bytes = uint32_t(w & ST_VALUE); // a single tile's count fits 32 bits
		if (have_prefix) {
			const int k = __builtin_ctzll(have_prefix);
			sum += wave_sum(lane < k ? bytes : 0u) + (read_lane(w, k) & ST_VALUE);
			exclusive = sum;
			return true;
		}
		sum += wave_sum(bytes);
		win -= WAVE;
	}
}


#endif // __HIPCC__

} // namespace oroch_hip
