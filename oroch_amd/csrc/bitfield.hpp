// bitfield.hpp -- one lane packs or unpacks one whole 16-byte block.
//
// bitpck.h:73-201 puts value k of a block at bit k * nbits of a 128-bit
// little-endian word.  With the width as a template parameter every field
// position is a compile-time constant, so a field costs one v_bfe_u32 (or one
// v_alignbit_b32 + v_and when it straddles two dwords) to read and one
// v_lshl_or_b32 (two when it straddles) to write -- no per-value division, no
// LDS atomics.  Lane j of a wavefront handles packed block j; the values move
// between this layout and the lane-major one (lane l owns values 4l..4l+3)
// through an LDS array of one slot per value.
#pragma once

#include <utility>

#include "common.hpp"

namespace oroch_hip {

#ifdef __HIPCC__

// N bits (1..32) at bit B of the 128-bit word w.
template <int B, int N>
__device__ __forceinline__ uint32_t field32(const uint32_t (&w)[4])
{
	static_assert(N >= 1 && N <= 32 && B >= 0 && B + N <= 128, "field outside the block");
	constexpr int d = B >> 5, s = B & 31;
	if constexpr (N == 32 && s == 0) {
		return w[d];
	} else if constexpr (s + N <= 32) {
		return __builtin_amdgcn_ubfe(w[d], uint32_t(s), uint32_t(N));
	} else {
		uint32_t x = __builtin_amdgcn_alignbit(w[d + 1], w[d], uint32_t(s));
		if constexpr (N < 32)
			x &= (1u << N) - 1u;
		return x;
	}
}

template <typename C, int NB, int K>
__device__ __forceinline__ C field(const uint32_t (&w)[4])
{
	constexpr int B = K * NB;
	if constexpr (NB <= 32) {
		return C(field32<B, NB>(w));
	} else {
		static_assert(sizeof(C) == 8, "wide fields need a 64-bit compute type");
		const uint32_t lo = field32<B, 32>(w);
		const uint32_t hi = field32<B + 32, NB - 32>(w);
		return C((uint64_t(hi) << 32) | lo);
	}
}

// Fields K0 .. K0+3 (as far as the block has them), then the next four.  The
// compiler barrier between groups keeps the scheduler from extracting a whole
// block into registers before storing any of it: a handful of live registers
// instead of one per field.
template <typename C, int NB, int K0>
__device__ __forceinline__ void unpack_from(const uint32_t (&w)[4], C *out)
{
	constexpr int CAP = 128 / NB;
	if constexpr (K0 < CAP) {
		out[K0] = field<C, NB, K0>(w);
		if constexpr (K0 + 1 < CAP)
			out[K0 + 1] = field<C, NB, K0 + 1>(w);
		if constexpr (K0 + 2 < CAP)
			out[K0 + 2] = field<C, NB, K0 + 2>(w);
		if constexpr (K0 + 3 < CAP)
			out[K0 + 3] = field<C, NB, K0 + 3>(w);
		if constexpr (K0 + 4 < CAP) {
			asm volatile("" ::: "memory");
			unpack_from<C, NB, K0 + 4>(w, out);
		}
	}
}

// All capacity(NB) fields of packed block j (in w) to slots[j * capacity ..).
// The slot address is formed here, where the capacity is a constant, so that
// the compiler can see its alignment and merge the stores.
template <typename C, int NB>
__device__ __forceinline__ void unpack_fields(const uint32_t (&w)[4], C *slots, int j)
{
	// j is made opaque so that the slot address is formed here and now: left
	// alone, the compiler hoists one address per width out of the caller's
	// block loop and keeps them all in registers.
	asm volatile("" : "+v"(j));
	unpack_from<C, NB, 0>(w, slots + j * (128 / NB));
}

// ORs the low N bits of x (already masked) in at bit B.
template <int B, int N>
__device__ __forceinline__ void deposit32(uint32_t x, uint32_t (&w)[4])
{
	static_assert(N >= 1 && N <= 32 && B >= 0 && B + N <= 128, "field outside the block");
	constexpr int d = B >> 5, s = B & 31;
	w[d] |= x << s;
	if constexpr (s + N > 32)
		w[d + 1] |= x >> (32 - s);
}

template <typename C, int NB, int K>
__device__ __forceinline__ void deposit(C u, uint32_t (&w)[4])
{
	constexpr int B = K * NB;
	if constexpr (NB <= 32) {
		deposit32<B, NB>(uint32_t(u), w);
	} else {
		deposit32<B, 32>(uint32_t(u), w);
		deposit32<B + 32, NB - 32>(uint32_t(uint64_t(u) >> 32), w);
	}
}

// MASKED: the values may have bits above NB (bitpfr packs the low bits of
// every value, bitpck.h:94); they are cut here.
template <typename C, int NB, bool MASKED>
__device__ __forceinline__ C low_bits(C u)
{
	if constexpr (MASKED && NB < int(sizeof(C)) * 8)
		return u & C((C(1) << NB) - 1);
	else
		return u;
}

template <typename C, int NB, bool MASKED, int K0>
__device__ __forceinline__ void pack_from(const C *in, uint32_t (&w)[4])
{
	constexpr int CAP = 128 / NB;
	if constexpr (K0 < CAP) {
		deposit<C, NB, K0>(low_bits<C, NB, MASKED>(in[K0]), w);
		if constexpr (K0 + 1 < CAP)
			deposit<C, NB, K0 + 1>(low_bits<C, NB, MASKED>(in[K0 + 1]), w);
		if constexpr (K0 + 2 < CAP)
			deposit<C, NB, K0 + 2>(low_bits<C, NB, MASKED>(in[K0 + 2]), w);
		if constexpr (K0 + 3 < CAP)
			deposit<C, NB, K0 + 3>(low_bits<C, NB, MASKED>(in[K0 + 3]), w);
		if constexpr (K0 + 4 < CAP) {
			asm volatile("" ::: "memory");
			pack_from<C, NB, MASKED, K0 + 4>(in, w);
		}
	}
}

// Packed block j from slots[j * capacity ..).
template <typename C, int NB, bool MASKED = false>
__device__ __forceinline__ void pack_fields(const C *slots, int j, uint32_t (&w)[4])
{
	w[0] = w[1] = w[2] = w[3] = 0;
	asm volatile("" : "+v"(j));
	pack_from<C, NB, MASKED, 0>(slots + j * (128 / NB), w);
}

// Widths below this keep the lane-per-value code: a block of such narrow
// fields holds 32..128 values and a single lane would spend longer on it than
// the whole wavefront does value by value.
constexpr int FIELDS_MIN_NB = 4;
// Slots past the block's 256 that the last packed block may spill into.
constexpr int FIELDS_PAD = 128 / FIELDS_MIN_NB;

// Binary decision tree over the (wave-uniform) width: five or six scalar
// branches lead to the code specialised for it.  The walk over the packed
// blocks (lane j takes blocks j, j + 64, ...) is inside each leaf: with the tree
// inside the walk the compiler evaluates every comparison of the tree ahead of
// the loop and parks the results in spilled scalar registers, some two hundred
// instructions a call.
template <typename C, int LO, int HI>
__device__ __forceinline__ void unpack_blocks_dispatch(uint32_t nb, const uint8_t *packed, uint32_t nblk,
						       int lane, C *slots)
{
	if constexpr (LO == HI) {
		for (uint32_t j = uint32_t(lane); j < nblk; j += 64u) {
			const uint2 *q = reinterpret_cast<const uint2 *>(packed) + j * 2;
			const uint2 a = q[0], b = q[1];
			const uint32_t w[4] = {a.x, a.y, b.x, b.y};
			unpack_fields<C, LO>(w, slots, int(j));
		}
	} else {
		constexpr int MID = (LO + HI) / 2;
		if (nb <= uint32_t(MID))
			unpack_blocks_dispatch<C, LO, MID>(nb, packed, nblk, lane, slots);
		else
			unpack_blocks_dispatch<C, MID + 1, HI>(nb, packed, nblk, lane, slots);
	}
}

#endif // __HIPCC__

} // namespace oroch_hip
