// Write bandwidth of three store patterns over one 1 GiB buffer (gfx950); a wavefront writes 2 KiB per step:
//   A: each lane writes 32 consecutive bytes as two 16-byte stores (lanes 32 bytes apart)
//   B: each lane writes 16 bytes per store, lanes 16 bytes apart (two stores 1 KiB apart)
//   C: as B, but a wavefront writes 4 KiB per step with four stores (the width of the 32-bit kernels' stores)
// and the same with a copy (read + write) instead of a fill.
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench_write tools/ubench_write.hip 2>/dev/null
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int PATTERN, bool COPY>
__global__ __launch_bounds__(256) void k_write(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t chunks,
						uint32_t seed)
{
	// a chunk is 2 KiB (4 KiB for pattern C)
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const size_t stride = size_t(gridDim.x) * 4;
	const uint4 fill = make_uint4(seed, seed + lane, seed ^ 5u, seed * 3u);
	for (size_t c = size_t(blockIdx.x) * 4 + wave; c < chunks; c += stride) {
		if (PATTERN == 0) {
			uint4 *p = dst + c * 128;
			const uint4 *q = src + c * 128;
			const uint4 a = COPY ? q[lane * 2] : fill, b = COPY ? q[lane * 2 + 1] : fill;
			p[lane * 2 + 0] = a;
			p[lane * 2 + 1] = b;
		} else if (PATTERN == 1) {
			uint4 *p = dst + c * 128;
			const uint4 *q = src + c * 128;
			const uint4 a = COPY ? q[lane] : fill, b = COPY ? q[lane + 64] : fill;
			p[lane] = a;
			p[lane + 64] = b;
		} else {
			uint4 *p = dst + c * 256;
			const uint4 *q = src + c * 256;
			uint4 a = fill, b = fill, d = fill, e = fill;
			if (COPY)
				a = q[lane], b = q[lane + 64], d = q[lane + 128], e = q[lane + 192];
			p[lane] = a;
			p[lane + 64] = b;
			p[lane + 128] = d;
			p[lane + 192] = e;
		}
	}
}

template <int PATTERN, bool COPY>
static float run(uint4 *dst, const uint4 *src, size_t bytes, int grid, hipEvent_t e0, hipEvent_t e1)
{
	const size_t chunks = bytes / (PATTERN == 2 ? 4096 : 2048);
	float best = 1e9f;
	for (int rep = 0; rep < 5; rep++) {
		hipEventRecord(e0);
		k_write<PATTERN, COPY><<<grid, 256>>>(dst, src, chunks, 7u + rep);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		best = ms < best ? ms : best;
	}
	return best;
}

int main()
{
	const size_t bytes = size_t(1) << 30;
	uint4 *dst, *src;
	if (hipMalloc(&dst, bytes) != hipSuccess || hipMalloc(&src, bytes) != hipSuccess)
		return 1;
	hipMemset(src, 1, bytes);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (int grid : {2048, 4096, 16384, 65536}) {
		const float f[3] = {run<0, false>(dst, src, bytes, grid, e0, e1), run<1, false>(dst, src, bytes, grid, e0, e1),
				    run<2, false>(dst, src, bytes, grid, e0, e1)};
		const float c[3] = {run<0, true>(dst, src, bytes, grid, e0, e1), run<1, true>(dst, src, bytes, grid, e0, e1),
				    run<2, true>(dst, src, bytes, grid, e0, e1)};
		for (int p = 0; p < 3; p++)
			printf("grid %6d pattern %c: fill %.4f ms, %.0f GB/s written; copy %.4f ms, %.0f GB/s read+written\n", grid,
			       "ABC"[p], f[p], bytes / (f[p] * 1e-3) / 1e9, c[p], 2 * bytes / (c[p] * 1e-3) / 1e9);
	}
	return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
