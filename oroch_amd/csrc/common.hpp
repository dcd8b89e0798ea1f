// common.hpp -- shared device helpers for the gfx950 kernels.
//
// Conventions.  A value of element type T (W bits) is carried as a raw W-bit
// pattern, zero-extended, in the compute type C (uint32_t for W <= 32,
// uint64_t for W = 64); arithmetic wraps modulo 2^W, which is what the
// reference's promoted-int arithmetic yields (integer_traits.h:31-59).
// Wavefronts are 64 lanes; one wavefront owns one block of values.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/oroch_hip.h"

namespace oroch_hip {

constexpr int WAVE = 64;

template <int WBITS, bool SGN>
struct elem
{
	static constexpr int W = WBITS;
	static constexpr int BYTES = WBITS / 8;
	static constexpr bool sgn = SGN;
	using C = std::conditional_t<(WBITS <= 32), uint32_t, uint64_t>;
	using S = std::conditional_t<WBITS == 8, uint8_t,
		  std::conditional_t<WBITS == 16, uint16_t,
		  std::conditional_t<WBITS == 32, uint32_t, uint64_t>>>;
	static constexpr C MASK = WBITS == 64 ? ~C(0) : C((uint64_t(1) << (WBITS % 64)) - 1);
	static constexpr C SIGN = C(1) << (WBITS - 1);
	// XOR-ing this in maps the type's own ordering onto unsigned ordering.
	static constexpr C FLIP = SGN ? SIGN : C(0);
};

// Runs `f(elem<..>{})` for the element type named by a dtype code.
template <typename F>
inline int dispatch_dtype(int dtype, F &&f)
{
	switch (dtype) {
#if defined(OROCH_ONLY_I32) // experiment builds: one element type compiles in a sixth of the time
	case OROCH_I32: return f(elem<32, true>{});
#elif defined(OROCH_ONLY_I64)
	case OROCH_I64: return f(elem<64, true>{});
#else
	case OROCH_I8: return f(elem<8, true>{});
	case OROCH_U8: return f(elem<8, false>{});
	case OROCH_I16: return f(elem<16, true>{});
	case OROCH_U16: return f(elem<16, false>{});
	case OROCH_I32: return f(elem<32, true>{});
	case OROCH_U32: return f(elem<32, false>{});
	case OROCH_I64: return f(elem<64, true>{});
	case OROCH_U64: return f(elem<64, false>{});
#endif
	}
	return OROCH_E_DTYPE;
}

// bitpck.h:54-57 -- values of a given width per 16-byte block.
__host__ __device__ constexpr uint32_t cap_of(uint32_t nbits)
{
	return 128u / nbits;
}

// Per-width constants: capacity and ceil(2^32 / capacity), so that
// floor(x / capacity) == umulhi(x, magic) for every x < 2^25.
struct width_table
{
	uint32_t cap[65];
	uint32_t magic[65];
	uint32_t magic16[65];  // ceil(2^16 / capacity): floor(x / cap) == (x * magic16) >> 16 for x < 512
};

constexpr width_table make_width_table()
{
	width_table t{};
	t.cap[0] = 1;
	t.magic[0] = 0;
	t.magic16[0] = 0;
	for (uint32_t b = 1; b <= 64; b++) {
		uint32_t c = 128u / b;
		t.cap[b] = c;
		t.magic[b] = uint32_t(((uint64_t(1) << 32) + c - 1) / c);
		t.magic16[b] = (65536u + c - 1) / c;
	}
	return t;
}

// Records a HIP event pair around the launches issued while it is alive, when
// profiling is armed (oroch_hip_profile_start); otherwise does nothing.
void profile_mark(int kernel_id, hipStream_t st, bool begin);
struct profile_scope
{
	int id;
	hipStream_t st;
	profile_scope(int kernel_id, hipStream_t s) : id(kernel_id), st(s)
	{
		profile_mark(id, st, true);
	}
	~profile_scope()
	{
		profile_mark(id, st, false);
	}
};

// Value of a run-time switch (OROCH_OPT_*); api.hip.
int option(int id);

// While one of these is alive on a thread, profile_scope records nothing there
// (recording creates events, which allocates).
struct no_profile_scope
{
	no_profile_scope();
	~no_profile_scope();
};

// Device address of the scratch array behind oroch_group_encode_static on the
// current device, looked up on the first call per device; group.hip.
int static_scratch_setup(void **ptr);

// oroch_group_decode / _find / _at for a table that is not trusted (read from a
// file or the network): every block stays inside the first stream_bytes bytes
// of src whatever the table says.  group.hip, array.hip.
int group_decode_within(int dtype, const void *src, const uint64_t *offsets, size_t n, void *dst,
			uint64_t stream_bytes, void *stream);
int group_find_within(int dtype, const void *src, const uint64_t *offsets, size_t n, const uint64_t *values,
		      size_t m, uint64_t *index, uint64_t stream_bytes, void *stream);
int group_at_within(int dtype, const void *src, const uint64_t *offsets, size_t n, const uint64_t *index,
		    size_t m, void *dst, uint64_t stream_bytes, void *stream);

// oroch_varint_decode that also stores the bytes the n values occupy to
// *consumed (device memory, may be null); varint.hip.
int varint_decode_counted(int dtype, const void *src, size_t nbytes, size_t n, void *dst, void *workspace,
			  size_t workspace_bytes, uint64_t *consumed, void *stream);

// Host-side check of a table of block offsets read from outside: rising,
// multiples of 16, none beyond stream_bytes.
bool offsets_table_sound(const uint64_t *table, size_t entries, uint64_t stream_bytes);

#ifdef __HIPCC__

__constant__ const width_table g_width = make_width_table();

__device__ __forceinline__ int lane_id()
{
	return int(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
}

// integer_traits.h:166-169 -- bit length, 0 for 0.
__device__ __forceinline__ uint32_t bitlen(uint32_t x)
{
	return 32u - uint32_t(__clz(x));
}
__device__ __forceinline__ uint32_t bitlen(uint64_t x)
{
	return 64u - uint32_t(__clzll(x));
}

// varint.h:56-59, 66-72 -- bytes of a varint holding a value of bit length nb.
__device__ __forceinline__ uint32_t vsp_bits(uint32_t nb)
{
	uint32_t s = (nb + 6u) / 7u;
	return s ? s : 1u;
}
template <typename C>
__device__ __forceinline__ uint32_t vsp(C u)
{
	return vsp_bits(bitlen(u));
}

// zigzag.h:47-55 on raw W-bit patterns; identity for unsigned element types.
template <typename E>
__device__ __forceinline__ typename E::C zz(typename E::C raw)
{
	using C = typename E::C;
	if constexpr (!E::sgn)
		return raw;
	else
		return ((raw << 1) ^ (C(0) - (raw >> (E::W - 1)))) & E::MASK;
}
template <typename E>
__device__ __forceinline__ typename E::C unzz(typename E::C u)
{
	using C = typename E::C;
	if constexpr (!E::sgn)
		return u;
	else
		return ((u >> 1) ^ (C(0) - (u & 1))) & E::MASK;
}

// The width table copied into LDS by each workgroup: {capacity, magic,
// magic16, space of 256 values} per width.  Kernels index it from LDS rather
// than from constant memory: a dynamically indexed LDS read costs one DS
// instruction, where a scalar load stalls the wave on the shared LDS/scalar
// counter.
__device__ __forceinline__ void load_width_table(uint4 *wt)
{
	for (int i = threadIdx.x; i < 65; i += blockDim.x) {
		const uint32_t c = g_width.cap[i];
		wt[i] = make_uint4(c, g_width.magic[i], g_width.magic16[i],
				   i ? 16u * ((256u + c - 1u) / c) : 0u);
	}
	__syncthreads();
}

// bitpck.h:61-71 -- 16 * ceil(n / capacity(nbits)); nbits in 1..64, n < 2^25.
__device__ __forceinline__ uint32_t pck_space(const uint4 *wt, uint32_t n, uint32_t nbits)
{
	const uint4 e = wt[nbits];
	return 16u * __umulhi(n + e.x - 1u, e.y);
}

// ---- wavefront primitives -------------------------------------------------

__device__ __forceinline__ uint32_t uniform(uint32_t x)
{
	return uint32_t(__builtin_amdgcn_readfirstlane(int(x)));
}
__device__ __forceinline__ uint64_t uniform(uint64_t x)
{
	uint32_t lo = uniform(uint32_t(x)), hi = uniform(uint32_t(x >> 32));
	return (uint64_t(hi) << 32) | lo;
}
__device__ __forceinline__ int uniform(int x)
{
	return __builtin_amdgcn_readfirstlane(x);
}

// Cross-lane moves by DPP (data-parallel primitives): a VALU operand modifier,
// a few cycles of latency against ~100 for a round trip through the LDS
// crossbar (ds_bpermute).  CTRL: 0xb1 / 0x4e quad swaps, 0x11n row_shr:n,
// 0x12n row_ror:n, 0x142 row_bcast:15, 0x143 row_bcast:31 (rows are 16 lanes).
// Lanes without a source (or masked out by RM, one bit per row) get `old`.
template <int CTRL, int RM = 0xf>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t old, uint32_t x)
{
	return uint32_t(__builtin_amdgcn_update_dpp(int(old), int(x), CTRL, RM, 0xf, false));
}
template <int CTRL, int RM = 0xf>
__device__ __forceinline__ int dpp_mov(int old, int x)
{
	return __builtin_amdgcn_update_dpp(old, x, CTRL, RM, 0xf, false);
}
template <int CTRL, int RM = 0xf>
__device__ __forceinline__ uint64_t dpp_mov(uint64_t old, uint64_t x)
{
	uint32_t lo = dpp_mov<CTRL, RM>(uint32_t(old), uint32_t(x));
	uint32_t hi = dpp_mov<CTRL, RM>(uint32_t(old >> 32), uint32_t(x >> 32));
	return (uint64_t(hi) << 32) | lo;
}

__device__ __forceinline__ uint32_t read_lane(uint32_t x, int l)
{
	return uint32_t(__builtin_amdgcn_readlane(int(x), l));
}
__device__ __forceinline__ int read_lane(int x, int l)
{
	return __builtin_amdgcn_readlane(x, l);
}
__device__ __forceinline__ uint64_t read_lane(uint64_t x, int l)
{
	return (uint64_t(read_lane(uint32_t(x >> 32), l)) << 32) | read_lane(uint32_t(x), l);
}

// Full-wave reduction; `ident` is the operation's identity.  The result is
// wave-uniform (read from lane 63).
template <typename V, typename Op>
__device__ __forceinline__ V wave_reduce(V x, V ident, Op op)
{
	x = op(x, dpp_mov<0xb1>(ident, x));
	x = op(x, dpp_mov<0x4e>(ident, x));
	x = op(x, dpp_mov<0x124>(ident, x));
	x = op(x, dpp_mov<0x128>(ident, x));
	x = op(x, dpp_mov<0x142, 0xa>(ident, x));
	x = op(x, dpp_mov<0x143, 0xc>(ident, x));
	return read_lane(x, WAVE - 1);
}
// Inclusive prefix over lanes 0..lane.
template <typename V, typename Op>
__device__ __forceinline__ V wave_scan(V x, V ident, Op op)
{
	x = op(x, dpp_mov<0x111>(ident, x));
	x = op(x, dpp_mov<0x112>(ident, x));
	x = op(x, dpp_mov<0x114>(ident, x));
	x = op(x, dpp_mov<0x118>(ident, x));
	x = op(x, dpp_mov<0x142, 0xa>(ident, x));
	x = op(x, dpp_mov<0x143, 0xc>(ident, x));
	return x;
}

template <typename V>
__device__ __forceinline__ V wave_min(V x)
{
	return wave_reduce(x, V(~V(0)), [](V a, V b) { return a < b ? a : b; });
}
template <typename V>
__device__ __forceinline__ V wave_max(V x)
{
	return wave_reduce(x, V(0), [](V a, V b) { return a > b ? a : b; });
}
template <typename V>
__device__ __forceinline__ V wave_sum(V x)
{
	return wave_reduce(x, V(0), [](V a, V b) { return V(a + b); });
}

// Inclusive prefix sum over lanes 0..lane.
__device__ __forceinline__ uint32_t wave_scan_sum(uint32_t x, int lane)
{
	(void) lane;
	return wave_scan(x, 0u, [](uint32_t a, uint32_t b) { return a + b; });
}
// Inclusive prefix maximum over lanes 0..lane; values are >= -1.
__device__ __forceinline__ int wave_scan_max(int x, int lane)
{
	(void) lane;
	return wave_scan(x, -1, [](int a, int b) { return a > b ? a : b; });
}
// Value of lane + o, or 0 past the last lane.
__device__ __forceinline__ uint32_t shfl_down_zero(uint32_t x, int o, int lane)
{
	uint32_t y = __shfl_down(x, o, WAVE);
	return lane + o < WAVE ? y : 0u;
}

// Orders this wavefront's LDS traffic: what lanes wrote before is visible to
// what any lane of the same wavefront reads after.  The hardware executes a
// wavefront's DS instructions in issue order, so the only thing to stop is the
// compiler moving accesses across this point.  (A fence builtin would also
// wait for every global load in flight, which defeats prefetching.)
__device__ __forceinline__ void wave_sync()
{
	asm volatile("" ::: "memory");
}

// ---- loading values ------------------------------------------------------------

// Sixteen consecutive elements at p (16-byte aligned) as raw W-bit patterns.
template <typename E>
__device__ __forceinline__ void load16(const typename E::S *p, typename E::C (&v)[16])
{
	const uint4 *q = reinterpret_cast<const uint4 *>(p);
	if constexpr (E::BYTES == 8) {
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const uint4 a = q[k];
			v[2 * k] = (uint64_t(a.y) << 32) | a.x;
			v[2 * k + 1] = (uint64_t(a.w) << 32) | a.z;
		}
	} else if constexpr (E::BYTES == 4) {
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const uint4 a = q[k];
			v[4 * k] = a.x, v[4 * k + 1] = a.y, v[4 * k + 2] = a.z, v[4 * k + 3] = a.w;
		}
	} else if constexpr (E::BYTES == 2) {
#pragma unroll
		for (int k = 0; k < 2; k++) {
			const uint4 a = q[k];
			const uint32_t d[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
			for (int i = 0; i < 4; i++) {
				v[8 * k + 2 * i] = d[i] & 0xffffu;
				v[8 * k + 2 * i + 1] = d[i] >> 16;
			}
		}
	} else {
		const uint4 a = q[0];
		const uint32_t d[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
		for (int i = 0; i < 4; i++) {
			v[4 * i] = d[i] & 0xffu;
			v[4 * i + 1] = (d[i] >> 8) & 0xffu;
			v[4 * i + 2] = (d[i] >> 16) & 0xffu;
			v[4 * i + 3] = d[i] >> 24;
		}
	}
}

// ---- bit-granular access to a byte image held in LDS -----------------------
//
// `img` is a 4-byte aligned LDS array; `bit` an absolute bit position in it.

// OR `nb` (<= 32) bits of `val` (already masked) into the image.
__device__ __forceinline__ void lds_or_bits(uint32_t *img, uint32_t bit, uint32_t val)
{
	uint32_t d = bit >> 5, sh = bit & 31u;
	uint64_t x = uint64_t(val) << sh;
	atomicOr(&img[d], uint32_t(x));
	uint32_t hi = uint32_t(x >> 32);
	if (hi)
		atomicOr(&img[d + 1], hi);
}
// OR up to 64 bits of `val` (already masked) into the image.
__device__ __forceinline__ void lds_or_bits(uint32_t *img, uint32_t bit, uint64_t val)
{
	uint32_t d = bit >> 5, sh = bit & 31u;
	uint64_t lo = val << sh;
	uint32_t top = sh ? uint32_t(val >> (64u - sh)) : 0u;
	uint32_t w0 = uint32_t(lo), w1 = uint32_t(lo >> 32);
	if (w0)
		atomicOr(&img[d], w0);
	if (w1)
		atomicOr(&img[d + 1], w1);
	if (top)
		atomicOr(&img[d + 2], top);
}

// Read `nb` (1..32) bits at an absolute bit position.
__device__ __forceinline__ uint32_t lds_get_bits32(const uint32_t *img, uint32_t bit, uint32_t nb)
{
	uint32_t d = bit >> 5, sh = bit & 31u;
	uint64_t x = (uint64_t(img[d + 1]) << 32) | img[d];
	uint32_t v = uint32_t(x >> sh);
	return nb >= 32 ? v : (v & ((1u << nb) - 1u));
}
// Read `nb` (1..64) bits at an absolute bit position.
__device__ __forceinline__ uint64_t lds_get_bits64(const uint32_t *img, uint32_t bit, uint32_t nb)
{
	uint32_t d = bit >> 5, sh = bit & 31u;
	uint64_t lo = (uint64_t(img[d + 1]) << 32) | img[d];
	uint64_t v = lo >> sh;
	if (sh)
		v |= uint64_t(img[d + 2]) << (64u - sh);
	return nb >= 64 ? v : (v & ((uint64_t(1) << nb) - 1u));
}
template <typename C>
__device__ __forceinline__ C lds_get_bits(const uint32_t *img, uint32_t bit, uint32_t nb)
{
	if constexpr (sizeof(C) == 4)
		return lds_get_bits32(img, bit, nb);
	else
		return lds_get_bits64(img, bit, nb);
}

#endif // __HIPCC__

} // namespace oroch_hip
