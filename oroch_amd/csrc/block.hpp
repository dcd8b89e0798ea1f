// block.hpp -- what the kernels that work on 256-value blocks share: the lane
// layout (one 64-lane wavefront per block, lane l holds values 4l .. 4l+3),
// loading and storing a block, decoding a block image, and the persistent loop
// that walks the block images of a stream.
//
// Reference behaviour reproduced here (paths below /root/reference/oroch):
//   integer_codec.h:153-214, 498-609       metadata bytes, decode dispatch, bitpfr lists
//   integer_group.h:62-76                  [meta][pad to 8][data] framing
//   bitpck.h:171-201, varint.h:85-115      16-byte block unpacking, varints
#pragma once

#include <type_traits>

#include "common.hpp"
#include "bitfield.hpp"

namespace oroch_hip {

constexpr int VPL = 4;                     // values per lane
constexpr int BLOCK = WAVE * VPL;          // values per block (256)
constexpr int WAVES_PER_WG = 4;

static_assert(BLOCK == OROCH_BLOCK_VALUES, "block size mismatch");
static_assert(sizeof(oroch_block_desc) == 32, "descriptor layout");

// min(x, cap) for a 64-bit x and a small cap, written with 32-bit compares:
// there is no scalar 64-bit less-than on this target, and the vector compare
// the compiler substitutes for a wave-uniform one has been seen paired with a
// scalar select on a stale condition bit.
__device__ __forceinline__ uint32_t clamp_u64(uint64_t x, uint32_t cap)
{
	const uint32_t lo = uint32_t(x), hi = uint32_t(x >> 32);
	return hi ? cap : (lo < cap ? lo : cap);
}
// min(a, b) of two wave-uniform 64-bit values, on 32-bit compares for the same reason.
__device__ __forceinline__ uint64_t min_u64(uint64_t a, uint64_t b)
{
	const uint32_t ah = uint32_t(a >> 32), bh = uint32_t(b >> 32);
	const bool less = ah < bh || (ah == bh && uint32_t(a) < uint32_t(b));
	return less ? a : b;
}
// Values in block number blk of an n-value array.
__device__ __forceinline__ int block_count(size_t n, size_t blk)
{
	return int(clamp_u64(uint64_t(n) - uint64_t(blk) * BLOCK, uint32_t(BLOCK)));
}

// Bytes of LDS one wavefront needs for its block image: the largest image is
// the raw copy (8 + 256 * sizeof T); the decoder also uses it as value
// scratch.  Three spare dwords let bit reads run past the last byte.
template <typename E>
constexpr int image_bytes()
{
	return ((8 + BLOCK * E::BYTES + 15) / 16) * 16 + 16;
}

// ---- loading a block --------------------------------------------------------

template <typename E>
__device__ __forceinline__ void load_block(const typename E::S *src, size_t base, int nv, int lane,
					   typename E::C (&v)[VPL])
{
	using S = typename E::S;
	using C = typename E::C;
	const S *p = src + base + size_t(lane) * VPL;
	if (lane * VPL + VPL <= nv) {
		// one vector load per lane: 4, 8, 16 or 32 bytes
		if constexpr (E::BYTES == 8) {
			const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(p);
			ulonglong2 a = q[0], b = q[1];
			v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
		} else if constexpr (E::BYTES == 4) {
			uint4 a = *reinterpret_cast<const uint4 *>(p);
			v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
		} else if constexpr (E::BYTES == 2) {
			uint2 a = *reinterpret_cast<const uint2 *>(p);
			v[0] = a.x & 0xffffu; v[1] = a.x >> 16; v[2] = a.y & 0xffffu; v[3] = a.y >> 16;
		} else {
			uint32_t a = *reinterpret_cast<const uint32_t *>(p);
			v[0] = a & 0xffu; v[1] = (a >> 8) & 0xffu; v[2] = (a >> 16) & 0xffu; v[3] = a >> 24;
		}
	} else {
#pragma unroll
		for (int s = 0; s < VPL; s++)
			v[s] = lane * VPL + s < nv ? C(p[s]) : C(0);
	}
}

template <typename E>
__device__ __forceinline__ void store_block(typename E::S *dst, size_t base, int nv, int lane,
					    const typename E::C (&v)[VPL])
{
	using S = typename E::S;
	S *p = dst + base + size_t(lane) * VPL;
	if (nv == BLOCK || lane * VPL + VPL <= nv) {
		if constexpr (E::BYTES == 8) {
			ulonglong2 *q = reinterpret_cast<ulonglong2 *>(p);
			q[0] = make_ulonglong2(v[0], v[1]);
			q[1] = make_ulonglong2(v[2], v[3]);
		} else if constexpr (E::BYTES == 4) {
			*reinterpret_cast<uint4 *>(p) = make_uint4(v[0], v[1], v[2], v[3]);
		} else if constexpr (E::BYTES == 2) {
			*reinterpret_cast<uint2 *>(p) = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
		} else {
			*reinterpret_cast<uint32_t *>(p) = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
		}
	} else {
#pragma unroll
		for (int s = 0; s < VPL; s++)
			if (lane * VPL + s < nv)
				p[s] = S(v[s]);
	}
}

// ---- decode -------------------------------------------------------------------------

// varint.h:85-115 at a byte position of an LDS image; advances pos.
__device__ __forceinline__ uint64_t get_varint(const uint8_t *img, uint32_t &pos)
{
	uint64_t value = 0;
	uint32_t shift = 0;
	for (;;) {
		uint32_t byte = img[pos++];
		if (shift < 64)
			value |= uint64_t(byte & 0x7fu) << shift;
		shift += 7;
		if ((byte & 0x80u) == 0 || shift > 70)
			break;
	}
	return value;
}

// Decodes `count` varints that start at byte `start` of the image, lanes
// working on 64 bytes at a time: a byte that ends a value (top bit clear)
// closes one; every byte ORs its 7 payload bits into its value's slot of
// `out` (LDS, zero filled by the caller).  Returns the bytes consumed.
//
// With `where` (LDS, one entry per value) value number i goes to
// out[where[i]] instead of out[i], shifted up by `shift` bits, and entries of
// `where` at or beyond `slots` are left out: the high parts of a patched
// block, OR-ed straight into the values they belong to.
template <typename C>
__device__ uint32_t varint_list_decode(const uint8_t *img8, uint32_t start, uint32_t limit,
				       uint32_t count, C *out, int lane, const uint32_t *where = nullptr,
				       uint32_t shift = 0, uint32_t slots = 0)
{
	uint32_t done = 0;   // values completed so far
	uint32_t carry = 0;  // bytes of the open value seen in earlier chunks
	uint32_t pos = start;
	while (done < count && pos < limit) {
		const uint32_t at = pos + uint32_t(lane);
		const uint32_t byte = at < limit ? img8[at] : 0u;
		const bool term = (byte & 0x80u) == 0;
		const uint64_t T = __ballot(term);
		const uint64_t below = T & ((uint64_t(1) << lane) - 1);
		const uint32_t r = uint32_t(__popcll(below));
		const int pt = below ? 63 - __clzll(below) : -1;
		const uint32_t k = uint32_t(lane - pt - 1) + (pt < 0 ? carry : 0u);
		const uint32_t vi = done + r;
		if (vi < count && k * 7u + shift < sizeof(C) * 8u) {
			C part = C(byte & 0x7fu) << (k * 7u + shift);
			const uint32_t to = where ? where[vi] : vi;
			if (part && (!where || to < slots)) {
				if constexpr (sizeof(C) == 8)
					atomicOr(reinterpret_cast<unsigned long long *>(&out[to]),
						 (unsigned long long) part);
				else
					atomicOr(&out[to], part);
			}
		}
		// how far does this chunk go: up to the terminator that completes
		// value `count`, else all 64 bytes
		const uint32_t nterm = uint32_t(__popcll(T));
		if (done + nterm >= count) {
			// position of the (count - done)-th terminator
			uint32_t need = count - done;
			uint64_t t = T;
			for (uint32_t i = 1; i < need; i++)
				t &= t - 1;
			pos += uint32_t(__builtin_ctzll(t)) + 1u;
			done = count;
		} else {
			const int last = T ? 63 - __clzll(T) : -1;
			carry = last < 0 ? carry + 64u : uint32_t(63 - last);
			done += nterm;
			pos += 64u;
		}
	}
	wave_sync();
	return pos - start;
}


// Byte p (0..15) of a 16-byte header held in two scalar registers.  (Plain
// variables, not an array: a dynamically indexed private array is moved to
// memory by the compiler and everything derived from it stops being scalar.)
__device__ __forceinline__ uint32_t hdr_byte(uint64_t h0, uint64_t h1, uint32_t p)
{
	const uint64_t w = (p & 8u) ? h1 : h0;
	return uint32_t(w >> ((p & 7u) * 8u)) & 0xffu;
}
// varint.h:85-115 from the header registers; advances p.  Without a loop over
// the bytes: the length comes from the first byte whose top bit is clear, and
// the 7-bit groups of the first eight bytes are closed up in three steps (to
// 14-bit groups in 16-bit lanes, 28 in 32, 56 in 64), a fixed two dozen scalar
// instructions where the loop took a dozen per byte.
__device__ __forceinline__ uint64_t hdr_varint(uint64_t h0, uint64_t h1, uint32_t &p)
{
	// lo: the eight bytes at p, hi: the bytes after them
	const uint32_t sh = (p & 15u) * 8u;
	uint64_t lo = h0, hi = h1;
	if (sh >= 64u) {
		lo = h1 >> (sh - 64u);
		hi = 0;
	} else if (sh) {
		lo = (h0 >> sh) | (h1 << (64u - sh));
		hi = h1 >> sh;
	}
	const uint64_t stop = ~lo & 0x8080808080808080ull;
	uint64_t tail = 0;
	uint32_t nbytes;
	if (stop) {
		const uint32_t bit = uint32_t(__builtin_ctzll(stop)); // 7, 15, ..., 63
		lo &= ~uint64_t(0) >> (63u - bit);
		nbytes = (bit + 1u) >> 3;
	} else {
		// nine bytes or more: bits 56..62 from the ninth, bit 63 from the tenth;
		// what a longer (malformed) one holds beyond is dropped
		tail = (hi & 0x7fu) << 56;
		nbytes = 9;
		if (hi & 0x80u) {
			tail |= (hi >> 8) << 63;
			nbytes = (hi & 0x8000u) ? 11 : 10;
		}
	}
	uint64_t y = lo & 0x7f7f7f7f7f7f7f7full;
	y = ((y & 0x7f007f007f007f00ull) >> 1) | (y & 0x007f007f007f007full);
	y = ((y & 0x3fff00003fff0000ull) >> 2) | (y & 0x00003fff00003fffull);
	y = ((y & 0x0fffffff00000000ull) >> 4) | (y & 0x000000000fffffffull);
	p += nbytes;
	return y | tail;
}

// Decodes the block whose image (len bytes) sits in `img`.  FULL: 256 values.
template <typename E, bool FULL>
__device__ __forceinline__ void decode_block(uint32_t *img, uint32_t len, int nv, int lane, typename E::C *val,
			     uint32_t *gp, const uint4 *wt, typename E::C (&v)[VPL])
{
	using C = typename E::C;
	uint8_t *img8 = reinterpret_cast<uint8_t *>(img);

	// metadata, integer_codec.h:153-175, 184-194, 206-214: the first 16 bytes,
	// read once and parsed in scalar registers
	const uint4 hw = *reinterpret_cast<const uint4 *>(img);
	const uint64_t h0 = (uint64_t(uniform(hw.y)) << 32) | uniform(hw.x);
	const uint64_t h1 = (uint64_t(uniform(hw.w)) << 32) | uniform(hw.z);
	uint32_t p = 1;
	const uint32_t enc = uint32_t(h0) & 0xffu;
	C origin = 0;
	uint32_t nb = 0, nout = 0, inb = 0, vnb = 0;
	if (enc == OROCH_NAUGHT || enc == OROCH_VARFOR || enc == OROCH_BITFOR || enc == OROCH_BITPFR)
		origin = unzz<E>(C(hdr_varint(h0, h1, p)) & E::MASK);
	if (enc == OROCH_BITPCK || enc == OROCH_BITFOR || enc == OROCH_BITPFR)
		nb = hdr_byte(h0, h1, p++);
	if (enc == OROCH_BITPFR) {
		nout = uint32_t(hdr_varint(h0, h1, p));
		inb = hdr_byte(h0, h1, p++);
		vnb = hdr_byte(h0, h1, p++);
	}
	const uint32_t doff = (p + 7u) & ~7u; // integer_group.h:70-72

#pragma unroll
	for (int s = 0; s < VPL; s++)
		v[s] = 0;

	// Reject what the encoder never produces instead of reading wild.
	const bool sane = (nb >= 1 && nb <= uint32_t(E::W)) || !(enc >= OROCH_BITPCK);
	if (enc == OROCH_NAUGHT) {
		// naught.h:42-47
#pragma unroll
		for (int s = 0; s < VPL; s++)
			v[s] = origin;
	} else if (enc == OROCH_NORMAL) {
		// normal.h:51-58
		const typename E::S *raw = reinterpret_cast<const typename E::S *>(img8 + doff);
#pragma unroll
		for (int s = 0; s < VPL; s++)
			if (FULL || lane * VPL + s < nv)
				v[s] = raw[lane * VPL + s];
	} else if ((enc == OROCH_BITPCK || enc == OROCH_BITFOR || enc == OROCH_BITPFR) && sane) {
		// bitpck.h:171-201.  Unpacked values go to val[] (one slot per value),
		// exceptions are OR-ed into their slots there, then every lane picks up
		// its four values.
		if (nb >= uint32_t(FIELDS_MIN_NB)) {
			// lane j unpacks packed block j with compile-time field positions
			const uint32_t nblk = pck_space(wt, uint32_t(nv), nb) >> 4;
			// (widths above 32 bits have more than 64 packed blocks)
			unpack_blocks_dispatch<C, FIELDS_MIN_NB, E::W>(nb, img8 + doff, nblk, lane, val);
		} else {
			// narrow fields: one value per lane and step
			const uint4 we = wt[nb];
			const uint32_t cap = uniform(we.x), m16 = uniform(we.z);
#pragma unroll
			for (int s = 0; s < VPL; s++) {
				const uint32_t pos = uint32_t(lane * VPL + s);
				const uint32_t blkn = (pos * m16) >> 16;
				const uint32_t k = pos - blkn * cap;
				C x = 0;
				if (FULL || pos < uint32_t(nv))
					x = lds_get_bits<C>(img, doff * 8u + blkn * 128u + k * nb, nb) & E::MASK;
				val[pos] = x;
			}
		}
		wave_sync();
		if (enc == OROCH_BITPFR && nout > 0 && nout <= uint32_t(nv) && vnb <= uint32_t(E::W) &&
		    inb <= 64) {
			// integer_codec.h:583-609, bitpfr.h:114-124
			const uint32_t ibase = doff + pck_space(wt, uint32_t(nv), nb);
			// a varint list of gaps is assembled by OR into zeroed LDS slots
			if (!inb) {
				for (uint32_t i = uint32_t(lane); i < nout; i += WAVE)
					gp[i] = 0;
				wave_sync();
			}
			const uint32_t ilen = inb ? pck_space(wt, nout, inb)
						  : varint_list_decode<uint32_t>(img8, ibase, len, nout, gp, lane);
			const uint32_t vbase = ibase + ilen;
			const uint4 iw = wt[inb ? inb : 1u], vw = wt[vnb ? vnb : 1u];
			const uint32_t icap = uniform(iw.x), imagic = uniform(iw.y);
			const uint32_t vcap = uniform(vw.x), vmagic = uniform(vw.y);
			// positions: running sum of gap + 1 (offset.h:62-67); every
			// exception ORs its high part into the slot of its position
			uint32_t carry = 0;
#pragma unroll
			for (int it = 0; it < VPL; it++) {
				if (uint32_t(it * WAVE) < nout) {
					const uint32_t e = uint32_t(it * WAVE + lane);
					uint32_t step = 0;
					if (e < nout) {
						if (inb) {
							const uint32_t bk = __umulhi(e, imagic);
							// (gaps inside a block fit 32 bits; wider fields only
							// in streams no encoder made)
							const uint32_t at = ibase * 8u + bk * 128u + (e - bk * icap) * inb;
							step = inb <= 32u ? lds_get_bits32(img, at, inb) : uint32_t(lds_get_bits64(img, at, inb));
						} else {
							step = gp[e];
						}
						step += 1u;
					}
					const uint32_t incl = wave_scan_sum(step, lane) + carry;
					carry = read_lane(incl, WAVE - 1);
					if (e < nout) {
						if (vnb) {
							if (incl - 1u < uint32_t(BLOCK)) {
								const uint32_t bk = __umulhi(e, vmagic);
								const C hi = lds_get_bits<C>(
									img, vbase * 8u + bk * 128u + (e - bk * vcap) * vnb, vnb);
								val[incl - 1u] |= C(hi << nb) & E::MASK;
							}
						} else {
							gp[e] = incl - 1u; // where the e-th high part goes
						}
					}
				}
			}
			if (!vnb) {
				// a varint list of high parts: every byte ORs its bits into the
				// value of its exception
				wave_sync();
				varint_list_decode<C>(img8, vbase, len, nout, val, lane, gp, nb, uint32_t(BLOCK));
			}
			wave_sync();
		}
		C u[VPL];
		if constexpr (sizeof(C) == 4) {
			const uint4 q = *reinterpret_cast<const uint4 *>(&val[lane * VPL]);
			u[0] = q.x, u[1] = q.y, u[2] = q.z, u[3] = q.w;
		} else {
#pragma unroll
			for (int s = 0; s < VPL; s++)
				u[s] = val[lane * VPL + s];
		}
#pragma unroll
		for (int s = 0; s < VPL; s++)
			v[s] = enc == OROCH_BITPCK ? unzz<E>(u[s]) : C((u[s] + origin) & E::MASK);
	} else if (enc == OROCH_VARINT || enc == OROCH_VARFOR) {
		// varint.h:141-147
		for (int i = lane; i < BLOCK; i += WAVE)
			val[i] = 0;
		wave_sync();
		varint_list_decode<C>(img8, doff, len, uint32_t(nv), val, lane);
#pragma unroll
		for (int s = 0; s < VPL; s++) {
			C u = val[lane * VPL + s] & E::MASK;
			v[s] = enc == OROCH_VARINT ? unzz<E>(u) : C((u + origin) & E::MASK);
		}
	}
}

// ---- walking the blocks of a stream ------------------------------------------------
//
// Wavefronts that walk blocks wave, wave + stride, ...  The stream offsets are
// fetched two blocks ahead and the block image one block ahead (into
// registers), so that the two dependent memory latencies of a block (offset,
// then bytes) overlap with the decoding of earlier blocks.
//
// Where the loads are waited for matters.  The compiler's wait counts assume
// the fewest younger memory operations any path can have issued, so a load
// under a condition, or stores after a load, turn the wait for that load into
// a wait for everything in flight.  All loads of the walk are therefore
// unconditional (clamped addresses: pieces past a block's length read its
// first piece again and are dropped), and the next block's bytes are awaited
// and moved to LDS once this block is decoded, ahead of its stores, where the
// loads for the block after that are issued too: what is younger than those
// loads at their wait are stores that have had a block's decoding to complete.
//
// want(blk, img, len) is asked once the image is in LDS: WALK_DECODE decodes the
// block and hands the values to sink(blk, nv, v), WALK_SKIP goes on to the
// next block, WALK_STOP ends this wavefront's walk.  All answers are wave-uniform.
constexpr int WALK_SKIP = 0, WALK_DECODE = 1, WALK_STOP = -1;

template <typename E, typename Want, typename Sink>
__device__ __forceinline__ void walk_blocks(const uint8_t *__restrict__ src,
					     const uint64_t *__restrict__ offsets, size_t n,
					     size_t nblocks, uint64_t stream_limit, Want want, Sink sink)
{
	using C = typename E::C;
	constexpr uint32_t MAXLEN = uint32_t(image_bytes<E>() - 16);
	constexpr int NCH = int((MAXLEN + 1023) / 1024); // 1 KiB pieces: one uint4 per lane each
	// image: the block's bytes; vals: one slot per value (list scratch, patches)
	__shared__ __attribute__((aligned(16))) uint32_t image[WAVES_PER_WG][image_bytes<E>() / 4];
	__shared__ __attribute__((aligned(16))) C vals[WAVES_PER_WG][BLOCK + FIELDS_PAD];
	__shared__ __attribute__((aligned(16))) uint32_t gaps[WAVES_PER_WG][BLOCK];
	__shared__ uint4 wt[65];
	const int wave = uniform(int(threadIdx.x / WAVE)), lane = lane_id();
	load_width_table(wt);
	const size_t stride = size_t(gridDim.x) * WAVES_PER_WG;
	uint32_t *img = image[wave];
	uint8_t *img8 = reinterpret_cast<uint8_t *>(img);

	// Offsets come from the caller's table and may be anything: every block is
	// confined to [0, offsets[nblocks]), the length of the stream, and to the
	// stream_limit of the caller (the size of the memory behind src when the
	// table comes from a file or the network), so a damaged table yields wrong
	// values but no read outside the stream.
	const uint64_t limit = min_u64(uniform(offsets[nblocks]), stream_limit);
	// a stream without a single 16-byte piece: the loads below go to the table
	// instead (it has two entries at least) and every block has length zero
	const uint8_t *bytes = limit >= 16 ? src : reinterpret_cast<const uint8_t *>(offsets);
	const uint64_t last_piece = limit >= 16 ? (limit - 16) & ~uint64_t(15) : 0;

	// Vector loads on purpose: a scalar load would be waited for at the next
	// LDS access (they share a counter), a vector load stays in flight.  Lane 0
	// gets offsets[blk], lane 1 offsets[blk + 1] (both the end of the stream for
	// blocks past the last).
	auto extent = [&](size_t blk, uint64_t &o) {
		size_t i = (blk < nblocks ? blk : nblocks) + size_t(lane & 1);
		i = i < nblocks ? i : nblocks;
		o = __builtin_nontemporal_load(offsets + i);
	};
	// (in the two lanes that hold the offsets, not on the scalar unit, which has
	// the header to parse and is the busier of the two in this loop: 64-bit
	// minima and comparisons take a dozen scalar instructions each)
	auto resolve = [&](uint64_t x, uint64_t &off, uint32_t &len) {
		uint64_t y = x < limit ? x : limit;      // lane 0: start, lane 1: end
		y = lane == 0 ? y & ~uint64_t(15) : y;
		const uint64_t start = dpp_mov<0x111>(uint64_t(0), y); // lane 1 sees lane 0's
		const uint64_t d = y >= start ? y - start : 0;
		const uint32_t dl = d < uint64_t(MAXLEN) ? uint32_t(d) : MAXLEN;
		off = read_lane(y, 0);
		len = read_lane(dl, 1);
	};
	// The bytes in flight, one 1 KiB piece per variable (named variables: as an
	// array they end up in scratch memory).
	static_assert(NCH >= 1 && NCH <= 3, "pieces of the longest image");
	uint4 im0, im1 = make_uint4(0, 0, 0, 0), im2 = make_uint4(0, 0, 0, 0);
	auto piece = [&](uint64_t off, uint32_t len, int c) {
		const uint32_t o = uint32_t(c) * 1024u + uint32_t(lane) * 16u;
		const uint64_t at = off + (o < len ? o : 0u);
		return *reinterpret_cast<const uint4 *>(bytes + (at < last_piece ? at : last_piece));
	};
	// (the second and third piece only for blocks that long: a branch on the
	// length, which is uniform, costs two scalar instructions, a piece some ten
	// vector ones and a load; the wait for the loads is for all of them at once
	// anyway, see above)
	auto fetch = [&](uint64_t off, uint32_t len) {
		im0 = piece(off, len, 0);
		if constexpr (NCH > 1) {
			if (len > 1024u)
				im1 = piece(off, len, 1);
		}
		if constexpr (NCH > 2) {
			if (len > 2048u)
				im2 = piece(off, len, 2);
		}
	};
	// All pieces go to LDS whatever the block's length, as far as the image has
	// room (what lies past the block is never looked at, and a store under a
	// condition that changes from block to block costs more scalar instructions
	// than the store).  The first 16 bytes are parsed as the header
	// whatever the length: zeros for a block without bytes.
	constexpr uint32_t ROOM = uint32_t(image_bytes<E>()); // of the image in LDS
	auto stage = [&](uint32_t len) {
		const bool none = len == 0;
		const uint32_t o = uint32_t(lane) * 16u;
		if (o < ROOM)
			*reinterpret_cast<uint4 *>(img8 + o) =
				make_uint4(none ? 0u : im0.x, none ? 0u : im0.y, none ? 0u : im0.z, none ? 0u : im0.w);
		if constexpr (NCH > 1) {
			if (len > 1024u && 1024u + o < ROOM)
				*reinterpret_cast<uint4 *>(img8 + 1024u + o) = im1;
		}
		if constexpr (NCH > 2) {
			if (len > 2048u && 2048u + o < ROOM)
				*reinterpret_cast<uint4 *>(img8 + 2048u + o) = im2;
		}
	};

	size_t blk = size_t(blockIdx.x) * WAVES_PER_WG + wave;
	// len0: the block staged in LDS; off1/len1 and im: the next block, in
	// flight; o1: the offsets of the one after it, in flight
	uint64_t o0, o1, off0, off1;
	uint32_t len0, len1;
	extent(blk, o0);
	extent(blk + stride, o1);
	resolve(o0, off0, len0);
	fetch(off0, len0);
	stage(len0);
	wave_sync();
	resolve(o1, off1, len1);
	fetch(off1, len1);
	extent(blk + 2 * stride, o1);
	// moves the next block in and sets the loads for the one after it going
	auto advance = [&]() {
		stage(len1);
		len0 = len1;
		resolve(o1, off1, len1);
		fetch(off1, len1);
		extent(blk + 3 * stride, o1);
	};
	// One block: false when the walk is to end.  FULL blocks are the loop's
	// business, where the number of stores per block is then fixed (see above);
	// a short last block is taken after the loop.
	auto step = [&](auto full) -> bool {
		constexpr bool FULL = decltype(full)::value;
		const int nv = FULL ? BLOCK : block_count(n, blk);
		const int w = want(blk, img, len0);
		if (w == WALK_STOP)
			return false;
		if (w == WALK_DECODE) {
			C v[VPL];
			decode_block<E, FULL>(img, len0, nv, lane, vals[wave], gaps[wave], wt, v);
			wave_sync();
			advance();
			sink(blk, nv, v);
		} else {
			wave_sync();
			advance();
		}
		wave_sync();
		return true;
	};
	const size_t nfull = n / BLOCK < nblocks ? n / BLOCK : nblocks;
	// (the first block apart: the loop is then entered, as it is repeated, with
	// a block's stores in flight behind the loads, and its waits for the loads
	// count on them)
	if (blk < nfull) {
		if (!step(std::true_type()))
			return;
		blk += stride;
	}
	for (; blk < nfull; blk += stride) {
		if (!step(std::true_type()))
			return;
	}
	if (blk < nblocks)
		step(std::false_type());
}

} // namespace oroch_hip
