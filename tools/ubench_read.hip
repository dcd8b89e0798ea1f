// Read bandwidth of two access patterns over one 1 GiB buffer (gfx950):
//   A: each lane takes 64 consecutive bytes as four 16-byte loads (lanes 64 bytes apart)
//   B: each lane takes 16 bytes per load, lanes 16 bytes apart (four loads 1 KiB apart)
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench_read tools/ubench_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int PATTERN>
__global__ __launch_bounds__(256) void k_read(const uint4 *__restrict__ src, size_t chunks, uint32_t *out)
{
	// a chunk is 4 KiB: what one wavefront reads per step
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const size_t stride = size_t(gridDim.x) * 4;
	uint32_t acc = 0;
	for (size_t c = size_t(blockIdx.x) * 4 + wave; c < chunks; c += stride) {
		const uint4 *p = src + c * 256;
		uint4 a, b, d, e;
		if (PATTERN == 0) {
			a = p[lane * 4 + 0]; b = p[lane * 4 + 1]; d = p[lane * 4 + 2]; e = p[lane * 4 + 3];
		} else {
			a = p[lane]; b = p[lane + 64]; d = p[lane + 128]; e = p[lane + 192];
		}
		acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ d.x ^ d.y ^ d.z ^ d.w ^ e.x ^ e.y ^ e.z ^ e.w;
	}
	if (acc == 0x12345678u)
		out[0] = acc;
}

int main()
{
	const size_t bytes = size_t(1) << 30;
	uint4 *src;
	uint32_t *out;
	if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess)
		return 1;
	hipMemset(src, 1, bytes);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const size_t chunks = bytes / 4096;
	for (int grid : {1024, 2048, 4096, 8192, 16384, 65536}) {
		for (int pattern = 0; pattern < 2; pattern++) {
			float best = 1e9f;
			for (int rep = 0; rep < 5; rep++) {
				hipEventRecord(e0);
				if (pattern == 0)
					k_read<0><<<grid, 256>>>(src, chunks, out);
				else
					k_read<1><<<grid, 256>>>(src, chunks, out);
				hipEventRecord(e1);
				hipEventSynchronize(e1);
				float ms;
				hipEventElapsedTime(&ms, e0, e1);
				best = ms < best ? ms : best;
			}
			printf("grid %6d pattern %c: %.4f ms, %.0f GB/s\n", grid, pattern ? 'B' : 'A', best, bytes / (best * 1e-3) / 1e9);
		}
	}
	return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
