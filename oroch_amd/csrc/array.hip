// array.hip -- searching and random access over a stream of block images, and the
// growable compressed array built on them.
//
// Reference behaviour reproduced here (paths below /root/reference/oroch):
//   integer_array.h:64-69      operator[] of a group: decode the block, pick one value
//   integer_array.h:71-136     find within a group, with the metadata width shortcut
//   integer_array.h:150-245    integer_array: size, at, find, clear, insert
//
// The kernels keep the mapping of the block decoder: one 64-lane wavefront per
// block, lane l holding values 4l .. 4l+3.
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "block.hpp"
#include "decode_quad.hpp"
#include "host_util.hpp"

namespace oroch_hip {

// Bits needed for x (0 for 0): integer_traits.h:44-60.
template <typename C>
__device__ __forceinline__ uint32_t used_bits(C x)
{
	if constexpr (sizeof(C) == 8)
		return x ? 64u - uint32_t(__clzll((unsigned long long) x)) : 0u;
	else
		return x ? 32u - uint32_t(__clz(uint32_t(x))) : 0u;
}

// integer_array.h:85-126: can the block whose image starts at img hold the
// value?  A bitpck block holds only values whose zigzag form fits nbits, a
// bitfor block only values within 2^nbits above its origin, a naught block
// only its origin.  Everything else has to be decoded to know.
template <typename E>
__device__ __forceinline__ bool may_hold(const uint32_t *img, typename E::C needle)
{
	using C = typename E::C;
	const uint4 hw = *reinterpret_cast<const uint4 *>(img);
	const uint64_t h0 = (uint64_t(uniform(hw.y)) << 32) | uniform(hw.x);
	const uint64_t h1 = (uint64_t(uniform(hw.w)) << 32) | uniform(hw.z);
	const uint32_t enc = uint32_t(h0) & 0xffu;
	uint32_t p = 1;
	if (enc == OROCH_BITPCK) {
		const uint32_t nb = hdr_byte(h0, h1, p);
		return used_bits<C>(zz<E>(needle)) <= nb;
	}
	if (enc == OROCH_NAUGHT || enc == OROCH_BITFOR) {
		const C origin = unzz<E>(C(hdr_varint(h0, h1, p)) & E::MASK);
		if (enc == OROCH_NAUGHT)
			return needle == origin;
		const uint32_t nb = hdr_byte(h0, h1, p);
		return used_bits<C>(C((needle - origin) & E::MASK)) <= nb;
	}
	return true;
}

// Has a position below block blk been reported already?  (Blocks are walked in
// rising order by every wavefront, so nothing it has left can improve on that.)
__device__ __forceinline__ bool found_below(const unsigned long long *result, size_t blk)
{
	const uint64_t best = read_lane(uint64_t(__atomic_load_n(result, __ATOMIC_RELAXED)), 0);
	const uint64_t here = uint64_t(blk) * BLOCK;
	// 32-bit compares: see clamp_u64
	const uint32_t bh = uint32_t(best >> 32), hh = uint32_t(here >> 32);
	return bh < hh || (bh == hh && uint32_t(best) < uint32_t(here));
}

// Compares the block's values with every needle; a lane that holds one
// reports its position by atomic minimum.
template <typename E>
__device__ __forceinline__ void report_hits(size_t blk, int nv, int lane, const typename E::C (&v)[VPL],
					    const uint64_t *__restrict__ needles, size_t m,
					    typename E::C first, unsigned long long *result)
{
	using C = typename E::C;
	if (m == 1) {
		// the one needle was read when the kernel started
		int hit = -1;
#pragma unroll
		for (int s = VPL - 1; s >= 0; s--)
			if (v[s] == first && lane * VPL + s < nv)
				hit = s;
		if (hit >= 0)
			atomicMin(&result[0], (unsigned long long) (uint64_t(blk) * BLOCK + uint64_t(lane * VPL + hit)));
		return;
	}
	// 64 needles per load, one per lane, handed round by lane reads
	for (size_t j0 = 0; j0 < m; j0 += WAVE) {
		const size_t left = m - j0;
		const int here = left < size_t(WAVE) ? int(left) : WAVE;
		const C mine = lane < here ? C(needles[j0 + size_t(lane)]) & E::MASK : C(0);
		for (int jj = 0; jj < here; jj++) {
			const C x = read_lane(mine, jj);
			// four compares per needle; the rest only where some lane has a match
			const bool any = v[0] == x || v[1] == x || v[2] == x || v[3] == x;
			if (__ballot(any) == 0)
				continue;
			int hit = -1;
#pragma unroll
			for (int s = VPL - 1; s >= 0; s--)
				if (v[s] == x && lane * VPL + s < nv)
					hit = s;
			if (hit >= 0)
				atomicMin(&result[j0 + size_t(jj)],
					  (unsigned long long) (uint64_t(blk) * BLOCK + uint64_t(lane * VPL + hit)));
		}
	}
}

// For each of m needles the lowest position holding it, by atomic minimum
// into result[] (preset to all ones = not found).  With one needle the
// wavefronts skip the blocks that cannot hold it and stop once a lower block
// has produced a hit.
template <typename E>
__global__ __launch_bounds__(WAVE *WAVES_PER_WG) void k_find(const uint8_t *__restrict__ src,
							       const uint64_t *__restrict__ offsets,
							       size_t n, size_t nblocks, uint64_t stream_limit,
							       const uint64_t *__restrict__ needles, size_t m,
							       unsigned long long *result)
{
	using C = typename E::C;
	const int lane = lane_id();
	// a workgroup that starts after a lower block has reported has nothing to do
	if (m == 1 && found_below(result, size_t(blockIdx.x) * WAVES_PER_WG))
		return;
	const C first = C(uniform(needles[0])) & E::MASK;
	walk_blocks<E>(
		src, offsets, n, nblocks, stream_limit,
		[&](size_t blk, const uint32_t *img, uint32_t) {
			if (m != 1)
				return WALK_DECODE;
			if (found_below(result, blk))
				return WALK_STOP;
			return may_hold<E>(img, first) ? WALK_DECODE : WALK_SKIP;
		},
		[&](size_t blk, int nv, const C(&v)[VPL]) { report_hits<E>(blk, nv, lane, v, needles, m, first, result); });
}

// The same on the four-blocks-per-wavefront decoder (element types up to 32
// bits); blocks are not skipped there, decoding four at once costs less than
// deciding.
template <typename E>
__global__ __launch_bounds__(WAVE *WAVES_PER_WG) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_find_quad(
	const uint8_t *__restrict__ src, const uint64_t *__restrict__ offsets, size_t n, size_t nblocks,
	uint64_t stream_limit, const uint64_t *__restrict__ needles, size_t m, unsigned long long *result)
{
	using C = typename E::C;
	// a workgroup that starts after a lower block has reported has nothing to do
	if (m == 1 && found_below(result, size_t(blockIdx.x) * WAVES_PER_WG * QUAD))
		return;
	const C first = C(uniform(needles[0])) & E::MASK;
	walk_quads<E>(
		src, offsets, n, nblocks, stream_limit,
		[&](size_t blk) { return m == 1 && found_below(result, blk) ? WALK_STOP : WALK_DECODE; },
		[&](size_t blk, int nv, int lane, const C(&v)[VPL]) {
			report_hits<E>(blk, nv, lane, v, needles, m, first, result);
		});
}

// ---- random access -------------------------------------------------------------------------

// Up to 64 bits at bit position `bit` of the packed 16-byte words that start
// at byte `at` of a block image (bitpck.h:203-225: a field never leaves its
// word); 0 where the word would reach beyond the image.
__device__ __forceinline__ uint64_t image_bits(const uint8_t *img, uint32_t len, uint32_t at, uint32_t bit,
					       uint32_t nbits)
{
	const uint32_t w = at + (bit >> 7) * 16u;
	if (w + 16u > len)
		return 0;
	const uint64_t *q = reinterpret_cast<const uint64_t *>(img + w);
	const uint32_t sh = bit & 127u;
	uint64_t x;
	if (sh < 64u) {
		x = q[0] >> sh;
		if (sh && sh + nbits > 64u)
			x |= q[1] << (64u - sh);
	} else {
		x = q[1] >> (sh - 64u);
	}
	return nbits >= 64u ? x : x & ((uint64_t(1) << nbits) - 1u);
}

// varint.h:85-115 at byte p of a block image in global memory; advances p.
__device__ __forceinline__ uint64_t image_varint(const uint8_t *img, uint32_t len, uint32_t &p)
{
	uint64_t value = 0;
	uint32_t shift = 0;
	for (;;) {
		const uint32_t byte = p < len ? img[p] : 0u;
		p++;
		if (shift < 64)
			value |= uint64_t(byte & 0x7fu) << shift;
		shift += 7;
		if ((byte & 0x80u) == 0 || shift > 70)
			break;
	}
	return value;
}

// One value of a block straight from its image in global memory, without
// decoding the block: what a fetch is for the bit-packed encodings
// (bitpck.h:203-225, 258-266), the constant and the raw ones.  A patched
// block (integer_codec.h:583-609) adds a walk over its exception positions,
// taken here for up to AT_WALK exceptions.  Returns false where the block has
// to be decoded instead (varints as values, long exception lists, metadata
// the encoder does not write).
constexpr uint32_t AT_WALK = 32;

template <typename E>
__device__ __forceinline__ bool value_direct(const uint8_t *img, uint32_t len, uint32_t nv, uint32_t pos,
					     const uint4 *wt, typename E::C &out)
{
	using C = typename E::C;
	out = 0;
	if (len < 16u)
		return true; // no image: decodes to zeros as well
	const uint4 hw = *reinterpret_cast<const uint4 *>(img);
	const uint64_t h0 = (uint64_t(hw.y) << 32) | hw.x, h1 = (uint64_t(hw.w) << 32) | hw.z;
	const uint32_t enc = uint32_t(h0) & 0xffu;
	uint32_t p = 1;
	if (enc == OROCH_NORMAL) {
		// normal.h:51-58
		const uint32_t at = 8u + pos * uint32_t(E::BYTES);
		if (at + uint32_t(E::BYTES) <= len)
			out = C(*reinterpret_cast<const typename E::S *>(img + at)) & E::MASK;
		return true;
	}
	if (enc != OROCH_NAUGHT && enc != OROCH_BITPCK && enc != OROCH_BITFOR && enc != OROCH_BITPFR)
		return false;
	C origin = 0;
	if (enc != OROCH_BITPCK)
		origin = unzz<E>(C(hdr_varint(h0, h1, p)) & E::MASK);
	if (enc == OROCH_NAUGHT) {
		out = origin; // naught.h:42-47
		return true;
	}
	const uint32_t nb = hdr_byte(h0, h1, p++);
	if (nb < 1u || nb > uint32_t(E::W))
		return false;
	uint32_t nout = 0, inb = 0, vnb = 0;
	if (enc == OROCH_BITPFR) {
		nout = uint32_t(hdr_varint(h0, h1, p));
		inb = hdr_byte(h0, h1, p++);
		vnb = hdr_byte(h0, h1, p++);
		if (nout > AT_WALK || nout > nv || vnb > uint32_t(E::W) || inb > 64u)
			return false;
	}
	const uint32_t doff = (p + 7u) & ~7u; // integer_group.h:70-72
	const uint4 w = wt[nb];
	const uint32_t word = __umulhi(pos, w.y); // pos / capacity
	C u = C(image_bits(img, len, doff, word * 128u + (pos - word * w.x) * nb, nb)) & E::MASK;
	if (nout) {
		// the position list, offset.h:62-67: each entry is the distance from
		// the entry before it, less one
		const uint32_t ibase = doff + pck_space(wt, nv, nb);
		const uint4 iw = wt[inb ? inb : 1u];
		uint32_t ip = ibase, next = 0, found = nout;
		for (uint32_t e = 0; e < nout; e++) {
			uint32_t gap;
			if (inb) {
				const uint32_t bk = __umulhi(e, iw.y);
				gap = uint32_t(image_bits(img, len, ibase, bk * 128u + (e - bk * iw.x) * inb, inb));
			} else {
				gap = uint32_t(image_varint(img, len, ip));
			}
			next += gap + 1u;
			if (next - 1u == pos)
				found = e;
			// (a list of varints is walked to its end: the values start there)
			if (inb && next > pos)
				break;
		}
		if (found < nout) {
			const uint32_t vbase = inb ? ibase + pck_space(wt, nout, inb) : ip;
			C hi;
			if (vnb) {
				const uint4 vw = wt[vnb];
				const uint32_t bk = __umulhi(found, vw.y);
				hi = C(image_bits(img, len, vbase, bk * 128u + (found - bk * vw.x) * vnb, vnb));
			} else {
				uint32_t vp = vbase;
				for (uint32_t e = 0; e < found; e++)
					(void) image_varint(img, len, vp);
				hi = C(image_varint(img, len, vp));
			}
			if (nb < uint32_t(E::W))
				u |= C(hi << nb) & E::MASK;
		}
	}
	out = enc == OROCH_BITPCK ? unzz<E>(u) : C((u + origin) & E::MASK);
	return true;
}

// Value number index[q] of the array for each of m queries.  Every lane takes
// one query and reads its value straight from the block image where the
// encoding allows; the queries that are left are then served one at a time by
// the whole wavefront, which decodes the block (array_integer_group<T>::
// operator[], integer_array.h:64-69).  A position at or beyond n yields 0.
template <typename E>
__global__ __launch_bounds__(WAVE *WAVES_PER_WG) void k_at(const uint8_t *__restrict__ src,
							     const uint64_t *__restrict__ offsets, size_t n,
							     size_t nblocks, uint64_t stream_limit,
							     const uint64_t *__restrict__ index, size_t m,
							     typename E::S *__restrict__ dst)
{
	using C = typename E::C;
	using S = typename E::S;
	constexpr uint32_t MAXLEN = uint32_t(image_bytes<E>() - 16);
	constexpr int NCH = int((MAXLEN + 1023) / 1024);
	__shared__ __attribute__((aligned(16))) uint32_t image[WAVES_PER_WG][image_bytes<E>() / 4];
	__shared__ __attribute__((aligned(16))) C vals[WAVES_PER_WG][BLOCK + FIELDS_PAD];
	__shared__ __attribute__((aligned(16))) uint32_t gaps[WAVES_PER_WG][BLOCK];
	__shared__ uint4 wt[65];
	const int wave = uniform(int(threadIdx.x / WAVE)), lane = lane_id();
	load_width_table(wt);
	uint32_t *img = image[wave];
	uint8_t *img8 = reinterpret_cast<uint8_t *>(img);
	const uint64_t limit = min_u64(uniform(offsets[nblocks]), stream_limit);
	const size_t stride = size_t(gridDim.x) * WAVES_PER_WG * WAVE;
	for (size_t q0 = (size_t(blockIdx.x) * WAVES_PER_WG + wave) * WAVE; q0 < m; q0 += stride) {
		const size_t q = q0 + size_t(lane);
		// my query: the block's extent, confined to the stream as in walk_blocks
		uint64_t o = 0;
		uint32_t len = 0, pos = 0, nv = 0;
		bool need = false;
		if (q < m) {
			const uint64_t at = index[q];
			C x = 0;
			if (at < uint64_t(n)) {
				const size_t blk = size_t(at / BLOCK);
				pos = uint32_t(at % BLOCK);
				const uint64_t left = uint64_t(n) - uint64_t(blk) * BLOCK;
				nv = left < uint64_t(BLOCK) ? uint32_t(left) : uint32_t(BLOCK);
				const uint64_t b = offsets[blk], e = offsets[blk + 1];
				o = (b < limit ? b : limit) & ~uint64_t(15);
				const uint64_t end = e < limit ? e : limit;
				len = end > o ? uint32_t(end - o < uint64_t(MAXLEN) ? end - o : uint64_t(MAXLEN)) : 0u;
				need = !value_direct<E>(src + o, len, nv, pos, wt, x);
			}
			if (!need)
				dst[q] = S(x);
		}
		// the queries that need their block decoded, one after the other
		for (uint64_t todo = __ballot(need); todo; todo &= todo - 1) {
			const int l = int(__builtin_ctzll(todo));
			const uint64_t bo = read_lane(o, l);
			const uint32_t blen = read_lane(len, l), bpos = read_lane(pos, l);
			const int bnv = int(read_lane(nv, l));
#pragma unroll
			for (int c = 0; c < NCH; c++) {
				const uint32_t b = uint32_t(c) * 1024u + uint32_t(lane) * 16u;
				uint4 x = make_uint4(0, 0, 0, 0);
				if (b < blen)
					x = *reinterpret_cast<const uint4 *>(src + bo + b);
				if (b < blen || b == 0)
					*reinterpret_cast<uint4 *>(img8 + b) = x;
			}
			wave_sync();
			C v[VPL];
			if (bnv == BLOCK)
				decode_block<E, true>(img, blen, bnv, lane, vals[wave], gaps[wave], wt, v);
			else
				decode_block<E, false>(img, blen, bnv, lane, vals[wave], gaps[wave], wt, v);
			if (uint32_t(lane) == bpos / VPL) {
				const uint32_t s = bpos % VPL;
				const C x = s == 0 ? v[0] : s == 1 ? v[1] : s == 2 ? v[2] : v[3];
				dst[q0 + size_t(l)] = S(x);
			}
			wave_sync();
		}
	}
}

// out[i] = in[i] + base: block offsets of a piece encoded at byte `base` of a
// longer stream.
__global__ void k_rebase(const uint64_t *in, size_t count, uint64_t base, uint64_t *out)
{
	for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count;
	     i += size_t(gridDim.x) * blockDim.x)
		out[i] = in[i] + base;
}

static unsigned walk_grid(size_t units)
{
	const size_t wgs = (units + WAVES_PER_WG - 1) / WAVES_PER_WG;
	return unsigned(wgs < 16384 ? wgs : 16384);
}

} // namespace oroch_hip

using namespace oroch_hip;

// oroch_group_find and oroch_group_at with every block confined to the first
// stream_bytes bytes of src (tables that come from a file or the network).
int oroch_hip::group_find_within(int dtype, const void *src, const uint64_t *offsets, size_t n,
				 const uint64_t *values, size_t m, uint64_t *index, uint64_t stream_bytes,
				 void *stream)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (m == 0)
		return 0;
	if (!values || !index || (n && (!src || !offsets)) || (reinterpret_cast<uintptr_t>(src) & 15u))
		return OROCH_E_ARGUMENT;
	hipStream_t st = static_cast<hipStream_t>(stream);
	int rc = int(hipMemsetAsync(index, 0xff, m * sizeof(uint64_t), st));
	const size_t nb = (n + BLOCK - 1) / BLOCK;
	if (rc || nb == 0)
		return rc;
	return dispatch_dtype(dtype, [&](auto e) {
		using E = decltype(e);
		if constexpr (sizeof(typename E::C) == 4) {
			// OROCH_OPT_FIND_ONE_BLOCK keeps to the one-block walker (tests)
			if (!option(OROCH_OPT_FIND_ONE_BLOCK)) {
				hipLaunchKernelGGL(k_find_quad<E>, dim3(walk_grid((nb + QUAD - 1) / QUAD)),
						   dim3(WAVE * WAVES_PER_WG), 0, st, static_cast<const uint8_t *>(src),
						   offsets, n, nb, stream_bytes, values, m,
						   reinterpret_cast<unsigned long long *>(index));
				return int(hipGetLastError());
			}
		}
		hipLaunchKernelGGL(k_find<E>, dim3(walk_grid(nb)), dim3(WAVE * WAVES_PER_WG), 0, st,
				   static_cast<const uint8_t *>(src), offsets, n, nb, stream_bytes, values, m,
				   reinterpret_cast<unsigned long long *>(index));
		return int(hipGetLastError());
	});
}

extern "C" int oroch_group_find(int dtype, const void *src, const uint64_t *offsets, size_t n,
				const uint64_t *values, size_t m, uint64_t *index, void *stream)
{
	return group_find_within(dtype, src, offsets, n, values, m, index, ~uint64_t(0), stream);
}

int oroch_hip::group_at_within(int dtype, const void *src, const uint64_t *offsets, size_t n,
			       const uint64_t *index, size_t m, void *dst, uint64_t stream_bytes, void *stream)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (m == 0)
		return 0;
	if (!index || !dst || !offsets || (n && !src) || (reinterpret_cast<uintptr_t>(src) & 15u))
		return OROCH_E_ARGUMENT;
	hipStream_t st = static_cast<hipStream_t>(stream);
	const size_t nb = (n + BLOCK - 1) / BLOCK;
	return dispatch_dtype(dtype, [&](auto e) {
		using E = decltype(e);
		hipLaunchKernelGGL(k_at<E>, dim3(walk_grid((m + WAVE - 1) / WAVE)), dim3(WAVE * WAVES_PER_WG), 0, st,
				   static_cast<const uint8_t *>(src), offsets, n, nb, stream_bytes, index, m,
				   static_cast<typename E::S *>(dst));
		return int(hipGetLastError());
	});
}

extern "C" int oroch_group_at(int dtype, const void *src, const uint64_t *offsets, size_t n,
			      const uint64_t *index, size_t m, void *dst, void *stream)
{
	return group_at_within(dtype, src, offsets, n, index, m, dst, ~uint64_t(0), stream);
}

// ---- the growable array -----------------------------------------------------------------
//
// Full blocks live in device memory as one stream of block images plus the
// table of their offsets; the last 0..255 values wait on the host until they
// fill a block (integer_array.h:255-258 keeps the same split, with one heap
// buffer per group).

struct oroch_array
{
	int dtype = 0;
	size_t nblocks = 0;      // block images in the stream
	size_t stream_bytes = 0; // = offsets[nblocks]
	dev_buf stream;
	size_t stream_cap = 0;   // bytes
	dev_buf offsets;
	size_t offsets_cap = 0;  // bytes
	dev_buf small;           // needles, positions and results of one query batch
	size_t small_cap = 0;    // bytes
	std::vector<uint8_t> tail; // raw values not yet in a block

	size_t width() const
	{
		return size_t(1) << (dtype >> 1);
	}
	size_t in_blocks() const
	{
		return nblocks * size_t(BLOCK);
	}
	size_t in_tail() const
	{
		return tail.size() / width();
	}
};

namespace {

// Makes buf hold at least `need` bytes, keeping the first `keep`.
int grow(dev_buf &buf, size_t &cap, size_t need, size_t keep)
{
	if (need <= cap)
		return 0;
	size_t want = cap * 2 > need ? cap * 2 : need;
	want = (want + 255) & ~size_t(255);
	dev_buf bigger;
	int rc = bigger.alloc(want);
	if (rc)
		return rc;
	if (keep && (rc = int(hipMemcpy(bigger.p, buf.p, keep, hipMemcpyDeviceToDevice))))
		return rc;
	buf.swap(bigger);
	cap = want;
	return 0;
}

int small_buffer(oroch_array *a, size_t bytes)
{
	return grow(a->small, a->small_cap, bytes, 0);
}

// Appends count values (a multiple of 256, device memory) as new blocks.
int encode_blocks(oroch_array *a, const void *d_values, size_t count)
{
	if (count == 0)
		return 0;
	const size_t nb = count / BLOCK;
	const size_t bound = oroch_group_bound(a->dtype, count);
	const size_t wsb = oroch_group_workspace(a->dtype, count);
	int rc;
	if ((rc = grow(a->stream, a->stream_cap, a->stream_bytes + bound, a->stream_bytes)) ||
	    (rc = grow(a->offsets, a->offsets_cap, (a->nblocks + nb + 1) * sizeof(uint64_t),
		       a->nblocks ? (a->nblocks + 1) * sizeof(uint64_t) : 0)))
		return rc;
	dev_buf piece, ws;
	if ((rc = piece.alloc((nb + 1) * sizeof(uint64_t))) || (rc = ws.alloc(wsb)))
		return rc;
	uint64_t total = 0;
	rc = oroch_group_encode(a->dtype, d_values, count, static_cast<uint8_t *>(a->stream.p) + a->stream_bytes,
				bound, static_cast<uint64_t *>(piece.p), ws.p, wsb, &total, nullptr);
	if (rc)
		return rc;
	const unsigned grid = unsigned((nb + 1 + 255) / 256 < 1024 ? (nb + 1 + 255) / 256 : 1024);
	hipLaunchKernelGGL(k_rebase, dim3(grid), dim3(256), 0, nullptr, static_cast<const uint64_t *>(piece.p),
			   nb + 1, uint64_t(a->stream_bytes), static_cast<uint64_t *>(a->offsets.p) + a->nblocks);
	if ((rc = int(hipGetLastError())) || (rc = int(hipStreamSynchronize(nullptr))))
		return rc;
	a->stream_bytes += total;
	a->nblocks += nb;
	return 0;
}

// Moves a full tail (256 values) into the stream.
int flush_tail(oroch_array *a)
{
	if (a->in_tail() < size_t(BLOCK))
		return 0;
	dev_buf in;
	int rc;
	if ((rc = in.alloc(a->tail.size())) ||
	    (rc = int(hipMemcpy(in.p, a->tail.data(), a->tail.size(), hipMemcpyHostToDevice))) ||
	    (rc = encode_blocks(a, in.p, BLOCK)))
		return rc;
	a->tail.clear();
	return 0;
}

int append_any(oroch_array *a, const void *src, size_t n, bool src_on_device)
{
	if (n == 0)
		return 0;
	const size_t w = a->width();
	const size_t held = a->in_tail();
	const size_t full = (held + n) / BLOCK * BLOCK; // values that go into blocks now
	const uint8_t *from = static_cast<const uint8_t *>(src);
	int rc = 0;
	size_t taken = 0; // values of src that went into blocks
	if (full) {
		taken = full - held;
		if (held == 0 && src_on_device) {
			rc = encode_blocks(a, src, full);
		} else {
			dev_buf in;
			if ((rc = in.alloc(full * w)))
				return rc;
			uint8_t *d = static_cast<uint8_t *>(in.p);
			if (held)
				rc = int(hipMemcpy(d, a->tail.data(), held * w, hipMemcpyHostToDevice));
			if (rc == 0)
				rc = src_on_device ? int(hipMemcpy(d + held * w, from, taken * w, hipMemcpyDeviceToDevice))
						   : staged_copy(d + held * w, from, taken * w, true, nullptr);
			if (rc == 0)
				rc = encode_blocks(a, d, full);
		}
		if (rc)
			return rc;
		a->tail.clear();
	}
	const size_t rest = n - taken; // < 256 once blocks were cut, else all of src
	if (rest) {
		const size_t at = a->tail.size();
		a->tail.resize(at + rest * w);
		if (src_on_device)
			rc = int(hipMemcpy(a->tail.data() + at, from + taken * w, rest * w, hipMemcpyDeviceToHost));
		else
			std::memcpy(a->tail.data() + at, from + taken * w, rest * w);
	}
	return rc;
}

uint64_t tail_value(const oroch_array *a, size_t i)
{
	uint64_t x = 0;
	std::memcpy(&x, a->tail.data() + i * a->width(), a->width()); // little endian host
	return x;
}

} // namespace

extern "C" int oroch_array_create(int dtype, oroch_array **out)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (!out)
		return OROCH_E_ARGUMENT;
	oroch_array *a = new (std::nothrow) oroch_array;
	if (!a)
		return int(hipErrorOutOfMemory);
	a->dtype = dtype;
	*out = a;
	return 0;
}

extern "C" void oroch_array_destroy(oroch_array *a)
{
	delete a;
}

extern "C" uint64_t oroch_array_size(const oroch_array *a)
{
	return a ? a->in_blocks() + a->in_tail() : 0;
}

extern "C" int oroch_array_clear(oroch_array *a)
{
	if (!a)
		return OROCH_E_ARGUMENT;
	a->nblocks = 0;
	a->stream_bytes = 0;
	a->tail.clear();
	return 0;
}

// (host allocations of the tail must not throw through the C boundary)
extern "C" int oroch_array_append(oroch_array *a, const void *src, size_t n)
{
	if (!a || (n && !src))
		return OROCH_E_ARGUMENT;
	try {
		return append_any(a, src, n, true);
	} catch (const std::bad_alloc &) {
		return int(hipErrorOutOfMemory);
	}
}

extern "C" int oroch_array_append_host(oroch_array *a, const void *src_host, size_t n)
{
	if (!a || (n && !src_host))
		return OROCH_E_ARGUMENT;
	try {
		return append_any(a, src_host, n, false);
	} catch (const std::bad_alloc &) {
		return int(hipErrorOutOfMemory);
	}
}

extern "C" int oroch_array_view(const oroch_array *a, const void **stream, const uint64_t **offsets,
				uint64_t *nblocks, uint64_t *stream_bytes, const void **tail_host,
				uint64_t *ntail)
{
	if (!a)
		return OROCH_E_ARGUMENT;
	if (stream)
		*stream = a->nblocks ? a->stream.p : nullptr;
	if (offsets)
		*offsets = a->nblocks ? static_cast<const uint64_t *>(a->offsets.p) : nullptr;
	if (nblocks)
		*nblocks = a->nblocks;
	if (stream_bytes)
		*stream_bytes = a->stream_bytes;
	if (tail_host)
		*tail_host = a->tail.data();
	if (ntail)
		*ntail = a->in_tail();
	return 0;
}

extern "C" int oroch_array_image(const oroch_array *a, uint64_t group, void *dst_host, size_t capacity,
				 uint64_t *bytes)
{
	if (!a || !bytes)
		return OROCH_E_ARGUMENT;
	if (group >= a->nblocks)
		return OROCH_E_RANGE;
	uint64_t ext[2];
	int rc = int(hipMemcpy(ext, static_cast<const uint64_t *>(a->offsets.p) + group, sizeof(ext),
			       hipMemcpyDeviceToHost));
	if (rc)
		return rc;
	if (ext[0] > ext[1] || ext[1] > a->stream_bytes)
		return OROCH_E_ARGUMENT;
	*bytes = ext[1] - ext[0];
	if (*bytes > capacity)
		return OROCH_E_CAPACITY;
	if (!dst_host)
		return OROCH_E_ARGUMENT;
	return int(hipMemcpy(dst_host, static_cast<const uint8_t *>(a->stream.p) + ext[0], size_t(*bytes),
			     hipMemcpyDeviceToHost));
}

// ---- the array as a container -----------------------------------------------------------
//
// The full groups are the container's first blocks as they are; the tail, if
// any, becomes its short last block.

extern "C" size_t oroch_array_save_bound(const oroch_array *a)
{
	if (!a)
		return 0;
	const size_t ntail = a->in_tail();
	return container_stream_offset(a->nblocks + (ntail ? 1 : 0)) + a->stream_bytes +
	       (ntail ? oroch_group_bound(a->dtype, ntail) : 0);
}

static int save_array(const oroch_array *a, void *dst_host, size_t capacity, uint64_t *total_host)
{
	const size_t ntail = a->in_tail();
	const size_t nb = a->nblocks + (ntail ? 1 : 0);
	// the tail's image
	std::vector<uint8_t> last(ntail ? oroch_group_bound(a->dtype, ntail) : 0);
	uint64_t last_offs[2] = {0, 0}, last_len = 0;
	int rc;
	if (ntail && (rc = oroch_group_encode_host(a->dtype, a->tail.data(), ntail, last.data(), last.size(),
						   last_offs, &last_len)))
		return rc;
	oroch_container_header h;
	std::memset(&h, 0, sizeof h);
	std::memcpy(h.magic, CONTAINER_MAGIC, sizeof h.magic);
	h.dtype = uint32_t(a->dtype);
	h.block_values = OROCH_BLOCK_VALUES;
	h.nvalues = a->in_blocks() + ntail;
	h.nblocks = nb;
	h.table_offset = sizeof h;
	h.stream_offset = container_stream_offset(nb);
	h.stream_bytes = a->stream_bytes + last_len;
	*total_host = h.stream_offset + h.stream_bytes;
	if (*total_host > capacity)
		return OROCH_E_CAPACITY;
	uint8_t *out = static_cast<uint8_t *>(dst_host);
	std::memcpy(out, &h, sizeof h);
	uint64_t *table = reinterpret_cast<uint64_t *>(out + h.table_offset);
	std::memset(table, 0, size_t(h.stream_offset - h.table_offset));
	if (a->nblocks) {
		if ((rc = int(hipMemcpy(table, a->offsets.p, (a->nblocks + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost))) ||
		    (rc = staged_copy(out + h.stream_offset, a->stream.p, a->stream_bytes, false, nullptr)))
			return rc;
	}
	table[a->nblocks] = a->stream_bytes;
	if (ntail) {
		table[nb] = a->stream_bytes + last_len;
		std::memcpy(out + h.stream_offset + a->stream_bytes, last.data(), size_t(last_len));
	}
	return 0;
}

extern "C" int oroch_array_save(const oroch_array *a, void *dst_host, size_t capacity, uint64_t *total_host)
{
	if (!a || !dst_host || !total_host)
		return OROCH_E_ARGUMENT;
	try {
		return save_array(a, dst_host, capacity, total_host);
	} catch (const std::bad_alloc &) {
		return int(hipErrorOutOfMemory);
	}
}

static int load_array(oroch_array *a, const void *src_host, size_t bytes)
{
	oroch_container_header h;
	if (bytes < sizeof h)
		return OROCH_E_ARGUMENT;
	std::memcpy(&h, src_host, sizeof h);
	int rc = oroch_container_check(&h, bytes);
	if (rc)
		return rc;
	if (int(h.dtype) != a->dtype)
		return OROCH_E_DTYPE;
	const uint8_t *in = static_cast<const uint8_t *>(src_host);
	const size_t ntail = size_t(h.nvalues % BLOCK), nfull = size_t(h.nblocks) - (ntail ? 1 : 0);
	const size_t w = a->width();
	// the table may be anything: every entry must be a multiple of 16 inside the
	// stream and none lower than the one before, the first 0, for reads to stay
	// inside the stream and appends to go on behind the last full group
	const uint64_t *table = reinterpret_cast<const uint64_t *>(in + h.table_offset);
	if ((size_t(h.nblocks) + 1) * sizeof(uint64_t) > h.stream_offset - h.table_offset ||
	    !offsets_table_sound(table, size_t(h.nblocks) + 1, h.stream_bytes))
		return OROCH_E_ARGUMENT;
	uint64_t first_offset = 0, full_bytes = 0;
	std::memcpy(&first_offset, table, sizeof first_offset);
	std::memcpy(&full_bytes, in + h.table_offset + nfull * sizeof(uint64_t), sizeof full_bytes);
	if (first_offset != 0)
		return OROCH_E_ARGUMENT;
	std::vector<uint8_t> tail(ntail * w);
	if (ntail && (rc = oroch_container_decode_host(src_host, bytes, nfull, 1, tail.data())))
		return rc;
	if (nfull) {
		// into buffers of their own, which take the place of the array's only
		// when both copies have arrived: a failure leaves the array as it was
		dev_buf stream, offsets;
		size_t stream_cap = 0, offsets_cap = 0;
		if ((rc = grow(stream, stream_cap, size_t(full_bytes) + 16, 0)) ||
		    (rc = grow(offsets, offsets_cap, (nfull + 1) * sizeof(uint64_t), 0)) ||
		    (rc = staged_copy(stream.p, in + h.stream_offset, size_t(full_bytes), true, nullptr)) ||
		    (rc = staged_copy(offsets.p, in + h.table_offset, (nfull + 1) * sizeof(uint64_t), true, nullptr)))
			return rc;
		a->stream.swap(stream);
		a->stream_cap = stream_cap;
		a->offsets.swap(offsets);
		a->offsets_cap = offsets_cap;
	}
	a->nblocks = nfull;
	a->stream_bytes = size_t(full_bytes);
	a->tail.swap(tail);
	return 0;
}

extern "C" int oroch_array_load(oroch_array *a, const void *src_host, size_t bytes)
{
	if (!a || !src_host)
		return OROCH_E_ARGUMENT;
	try {
		return load_array(a, src_host, bytes);
	} catch (const std::bad_alloc &) {
		return int(hipErrorOutOfMemory);
	}
}

extern "C" int oroch_array_find(oroch_array *a, const uint64_t *values, size_t m, uint64_t *index)
{
	if (!a || (m && (!values || !index)))
		return OROCH_E_ARGUMENT;
	if (m == 0)
		return 0;
	for (size_t j = 0; j < m; j++)
		index[j] = ~uint64_t(0);
	int rc;
	if (a->nblocks) {
		if ((rc = small_buffer(a, 2 * m * sizeof(uint64_t))))
			return rc;
		uint64_t *d_values = static_cast<uint64_t *>(a->small.p), *d_index = d_values + m;
		if ((rc = int(hipMemcpy(d_values, values, m * sizeof(uint64_t), hipMemcpyHostToDevice))) ||
		    (rc = group_find_within(a->dtype, a->stream.p, static_cast<const uint64_t *>(a->offsets.p),
					    a->in_blocks(), d_values, m, d_index, a->stream_bytes, nullptr)) ||
		    (rc = int(hipMemcpy(index, d_index, m * sizeof(uint64_t), hipMemcpyDeviceToHost))))
			return rc;
	}
	// integer_array.h:202-205: the tail is searched last
	const size_t held = a->in_tail();
	const uint64_t mask = a->width() == 8 ? ~uint64_t(0) : (uint64_t(1) << (8 * a->width())) - 1;
	for (size_t j = 0; j < m && held; j++) {
		if (index[j] != ~uint64_t(0))
			continue;
		for (size_t i = 0; i < held; i++) {
			if (tail_value(a, i) == (values[j] & mask)) {
				index[j] = a->in_blocks() + i;
				break;
			}
		}
	}
	return 0;
}

extern "C" int oroch_array_at(oroch_array *a, const uint64_t *index, size_t m, void *dst_host)
{
	if (!a || (m && (!index || !dst_host)))
		return OROCH_E_ARGUMENT;
	const size_t size = a->in_blocks() + a->in_tail(), w = a->width();
	bool any_block = false;
	for (size_t q = 0; q < m; q++) {
		if (index[q] >= size)
			return OROCH_E_RANGE;
		any_block |= index[q] < a->in_blocks();
	}
	int rc;
	if (any_block) {
		const size_t ibytes = m * sizeof(uint64_t);
		if ((rc = small_buffer(a, ibytes + m * w)))
			return rc;
		uint64_t *d_index = static_cast<uint64_t *>(a->small.p);
		void *d_out = static_cast<uint8_t *>(a->small.p) + ibytes;
		if ((rc = int(hipMemcpy(d_index, index, ibytes, hipMemcpyHostToDevice))) ||
		    (rc = group_at_within(a->dtype, a->stream.p, static_cast<const uint64_t *>(a->offsets.p),
					  a->in_blocks(), d_index, m, d_out, a->stream_bytes, nullptr)) ||
		    (rc = int(hipMemcpy(dst_host, d_out, m * w, hipMemcpyDeviceToHost))))
			return rc;
	}
	for (size_t q = 0; q < m; q++) {
		if (index[q] >= a->in_blocks())
			std::memcpy(static_cast<uint8_t *>(dst_host) + q * w,
				    a->tail.data() + (index[q] - a->in_blocks()) * w, w);
	}
	return 0;
}

extern "C" int oroch_array_read(oroch_array *a, uint64_t first, uint64_t count, void *dst_host)
{
	if (!a || (count && !dst_host))
		return OROCH_E_ARGUMENT;
	const size_t size = a->in_blocks() + a->in_tail(), w = a->width();
	if (first > size || count > size - first)
		return OROCH_E_RANGE;
	uint8_t *out = static_cast<uint8_t *>(dst_host);
	const uint64_t stop = first + count;
	const uint64_t bstop = stop < a->in_blocks() ? stop : a->in_blocks();
	if (first < bstop) {
		const size_t b0 = first / BLOCK, b1 = (bstop + BLOCK - 1) / BLOCK;
		const size_t nv = (b1 - b0) * BLOCK;
		dev_buf vals;
		int rc;
		if ((rc = vals.alloc(nv * w)) ||
		    (rc = group_decode_within(a->dtype, a->stream.p, static_cast<const uint64_t *>(a->offsets.p) + b0,
					      nv, vals.p, a->stream_bytes, nullptr)) ||
		    (rc = staged_copy(out, static_cast<uint8_t *>(vals.p) + (first - b0 * BLOCK) * w,
				      (bstop - first) * w, false, nullptr)))
			return rc;
	}
	if (stop > a->in_blocks()) {
		const uint64_t t0 = first > a->in_blocks() ? first - a->in_blocks() : 0;
		const uint64_t skip = first > a->in_blocks() ? 0 : a->in_blocks() - first;
		std::memcpy(out + skip * w, a->tail.data() + t0 * w, (stop - a->in_blocks() - t0) * w);
	}
	return 0;
}

// integer_array.h:216-245.  An insertion below the tail moves every later
// value up by one: the blocks from the insertion point on are decoded, shifted
// on the device and encoded again; the value pushed out of the last block
// becomes the first of the tail.
static int insert_value(oroch_array *a, uint64_t index, uint64_t value)
{
	if (!a)
		return OROCH_E_ARGUMENT;
	const size_t w = a->width();
	if (index > a->in_blocks() + a->in_tail())
		return OROCH_E_RANGE;
	uint8_t raw[8];
	std::memcpy(raw, &value, 8); // little endian host
	int rc;
	if (index >= a->in_blocks()) {
		a->tail.insert(a->tail.begin() + (index - a->in_blocks()) * w, raw, raw + w);
		return flush_tail(a);
	}
	const size_t g = index / BLOCK, at = index - g * BLOCK;
	const size_t cnt = (a->nblocks - g) * BLOCK;
	dev_buf vals, moved;
	uint64_t cut = 0;
	if ((rc = vals.alloc(cnt * w)) || (rc = moved.alloc((cnt + 1) * w)) ||
	    (rc = group_decode_within(a->dtype, a->stream.p, static_cast<const uint64_t *>(a->offsets.p) + g, cnt,
				      vals.p, a->stream_bytes, nullptr)) ||
	    (rc = int(hipMemcpy(&cut, static_cast<const uint64_t *>(a->offsets.p) + g, sizeof(cut),
				hipMemcpyDeviceToHost))))
		return rc;
	uint8_t *v = static_cast<uint8_t *>(vals.p), *mv = static_cast<uint8_t *>(moved.p);
	uint8_t last[8];
	if ((at && (rc = int(hipMemcpy(mv, v, at * w, hipMemcpyDeviceToDevice)))) ||
	    (rc = int(hipMemcpy(mv + at * w, raw, w, hipMemcpyHostToDevice))) ||
	    (rc = int(hipMemcpy(mv + (at + 1) * w, v + at * w, (cnt - at) * w, hipMemcpyDeviceToDevice))) ||
	    (rc = int(hipMemcpy(last, mv + cnt * w, w, hipMemcpyDeviceToHost))))
		return rc;
	// The new images and their offsets (rebased onto the stream) are made beside
	// the old ones; every allocation, kernel and synchronisation is over before
	// the array is touched.  What then remains are two device-to-device copies
	// into memory the array already owns; short of losing the device they do not
	// fail, and a failure of anything before them leaves the array as it was.
	const size_t nb = cnt / BLOCK;
	const size_t bound = oroch_group_bound(a->dtype, cnt), wsb = oroch_group_workspace(a->dtype, cnt);
	dev_buf images, piece, ws;
	uint64_t total = 0;
	if ((rc = images.alloc(bound)) || (rc = piece.alloc((nb + 1) * sizeof(uint64_t))) || (rc = ws.alloc(wsb)) ||
	    (rc = oroch_group_encode(a->dtype, mv, cnt, images.p, bound, static_cast<uint64_t *>(piece.p), ws.p, wsb,
				     &total, nullptr)) ||
	    (rc = grow(a->stream, a->stream_cap, size_t(cut) + size_t(total) + 16, a->stream_bytes)))
		return rc;
	a->tail.reserve(a->tail.size() + w); // may throw, before anything has changed
	const unsigned grid = unsigned((nb + 1 + 255) / 256 < 1024 ? (nb + 1 + 255) / 256 : 1024);
	hipLaunchKernelGGL(k_rebase, dim3(grid), dim3(256), 0, nullptr, static_cast<const uint64_t *>(piece.p), nb + 1,
			   cut, static_cast<uint64_t *>(piece.p));
	if ((rc = int(hipGetLastError())) || (rc = int(hipStreamSynchronize(nullptr))))
		return rc;
	if ((rc = int(hipMemcpy(static_cast<uint8_t *>(a->stream.p) + cut, images.p, size_t(total),
				hipMemcpyDeviceToDevice))) ||
	    (rc = int(hipMemcpy(static_cast<uint64_t *>(a->offsets.p) + g, piece.p, (nb + 1) * sizeof(uint64_t),
				hipMemcpyDeviceToDevice))))
		return rc;
	a->stream_bytes = size_t(cut) + size_t(total);
	a->tail.insert(a->tail.begin(), last, last + w);
	return flush_tail(a);
}

extern "C" int oroch_array_insert(oroch_array *a, uint64_t index, uint64_t value)
{
	try {
		return insert_value(a, index, value);
	} catch (const std::bad_alloc &) {
		return int(hipErrorOutOfMemory);
	}
}
