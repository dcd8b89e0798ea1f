// decode_quad.hpp -- the decoder's fast path: four consecutive blocks per
// wavefront, sixteen lanes per block.
//
// With one block per wavefront most of the per-block work (header parsing,
// exception lists) is wave-uniform and ends up on the scalar unit, and the
// bit-field extraction keeps only as many lanes busy as the block has packed
// 16-byte words (22 of 64 at 10 bits).  Here four blocks share a wavefront:
// each 16-lane group parses its own block's header and exception lists in
// vector registers (DPP row operations stay inside a group), and the packed
// words of all four blocks are spread over all 64 lanes.  The path covers what
// the selector produces for ordinary numeric data -- bitpck, bitfor and bitpfr
// blocks of one common width, exception lists coded as varints -- and anything
// else (other encodings, mixed widths, bit-packed or long lists, a short last
// block) goes block by block through decode_block, from the same staged bytes.
//
// Needs block.hpp; 8-, 16- and 32-bit element types.
#pragma once

namespace oroch_hip {

constexpr int QUAD = 4;            // blocks per wavefront step
constexpr int GROUP = WAVE / QUAD; // lanes per block
constexpr int SLOT_STRIDE = BLOCK + FIELDS_PAD;
constexpr int LIST_BYTES = GROUP * 4; // exception-list bytes a group looks at
constexpr int ENDS = 48;              // varint end positions per group (+ dump slots)
constexpr uint32_t QUAD_MAX_EXC = 16; // one exception per lane of a group

// Inclusive prefix sum within each row of 16 lanes.
__device__ __forceinline__ uint32_t row_scan_sum(uint32_t x)
{
	x += dpp_mov<0x111>(0u, x);
	x += dpp_mov<0x112>(0u, x);
	x += dpp_mov<0x114>(0u, x);
	x += dpp_mov<0x118>(0u, x);
	return x;
}

__device__ __forceinline__ uint32_t lane_gather(uint32_t x, int src_lane)
{
	return uint32_t(__builtin_amdgcn_ds_bpermute(src_lane * 4, int(x)));
}

// The 7-bit groups of the four bytes in lo up to and including the first byte
// without a continuation bit (varint.h:85-115); *more is set when all four
// bytes continue, i.e. a fifth byte belongs to the value.
__device__ __forceinline__ uint32_t varint4(uint32_t lo, bool *more, uint32_t *nbytes)
{
	const uint32_t t = ~lo & 0x80808080u;
	const uint32_t keep = t ^ (t - 1u); // bits up to the terminator's top bit; all ones if none
	lo &= keep;
	*more = t == 0;
	*nbytes = t ? (uint32_t(__builtin_ctz(t)) >> 3) + 1u : 5u;
	return (lo & 0x7fu) | ((lo >> 1) & 0x3f80u) | ((lo >> 2) & 0x1fc000u) | ((lo >> 3) & 0xfe00000u);
}

// The varint (at most five bytes) that starts at byte `at` of an LDS image.
__device__ __forceinline__ uint32_t varint_at(const uint32_t *img, uint32_t at)
{
	const uint32_t d = at >> 2, s = (at & 3u) * 8u;
	const uint32_t a = img[d], b = img[d + 1];
	const uint32_t lo = __builtin_amdgcn_alignbit(b, a, s);
	bool more;
	uint32_t nbytes;
	uint32_t v = varint4(lo, &more, &nbytes);
	if (more) {
		// byte 4 of the value sits 32 bits above `at`
		const uint32_t c = img[d + 2];
		const uint32_t hi = __builtin_amdgcn_alignbit(c, b, s);
		v |= (hi & 0x0fu) << 28;
	}
	return v;
}

// Per-block metadata as each lane of the block's group sees it.
struct quad_header
{
	uint32_t enc, nb, origin, nout, inb, vnb, doff;
	bool plain; // parsed completely by the short-form rules below
};

// integer_codec.h:153-214 in vector registers, for the three bit-packed
// encodings of element types up to 32 bits: [enc][origin varint <= 5 bytes]
// [nbits][noutliers varint <= 2 bytes][index nbits][value nbits].
template <typename E>
__device__ __forceinline__ quad_header parse_header(const uint4 h)
{
	quad_header r;
	r.enc = h.x & 0xffu;
	const bool has_origin = r.enc == OROCH_BITFOR || r.enc == OROCH_BITPFR;
	// bytes 1..12 of the header
	const uint32_t y0 = __builtin_amdgcn_alignbyte(h.y, h.x, 1);
	const uint32_t y1 = __builtin_amdgcn_alignbyte(h.z, h.y, 1);
	const uint32_t y2 = __builtin_amdgcn_alignbyte(h.w, h.z, 1);
	bool more;
	uint32_t olen;
	uint32_t zo = varint4(y0, &more, &olen);
	if (more)
		zo |= (y1 & 0x0fu) << 28;
	if (!has_origin) {
		zo = 0;
		olen = 0;
	}
	r.origin = unzz<E>(zo & E::MASK);
	// the bytes after the origin
	const bool far = olen >= 4;
	const uint32_t a = far ? y1 : y0, b = far ? y2 : y1, c = far ? 0u : y2;
	const uint32_t sh = (olen & 3u) * 8u;
	const uint32_t z0 = __builtin_amdgcn_alignbit(b, a, sh);
	const uint32_t z1 = __builtin_amdgcn_alignbit(c, b, sh);
	r.nb = z0 & 0xffu;
	const uint32_t b1 = (z0 >> 8) & 0xffu, b2 = (z0 >> 16) & 0xffu;
	const bool two = (b1 & 0x80u) != 0;
	r.plain = r.enc == OROCH_BITPCK || has_origin;
	uint32_t p = 1u + olen + 1u;
	r.nout = 0;
	r.inb = 0;
	r.vnb = 0;
	if (r.enc == OROCH_BITPFR) {
		r.nout = (b1 & 0x7fu) | (two ? (b2 & 0x7fu) << 7 : 0u);
		r.inb = two ? z0 >> 24 : b2;
		r.vnb = two ? z1 & 0xffu : z0 >> 24;
		r.plain = r.plain && !(two && (b2 & 0x80u));
		p += two ? 4u : 3u;
	}
	r.doff = (p + 7u) & ~7u; // integer_group.h:70-72
	return r;
}

// The blocks of a quad that share a width form a class (a 4-bit mask, uniform);
// rank_to_group() lists its members in ascending order.
struct quad_class
{
	uint32_t count;   // blocks in the class
	uint32_t member0, member1, member2, member3;
};

__device__ __forceinline__ quad_class make_class(uint32_t mask)
{
	quad_class c;
	c.count = uint32_t(__builtin_popcount(mask));
	uint32_t m = mask | 0x10u;
	c.member0 = uint32_t(__builtin_ctz(m)) & 3u;
	m = (m & (m - 1u)) | 0x10u;
	c.member1 = uint32_t(__builtin_ctz(m)) & 3u;
	m = (m & (m - 1u)) | 0x10u;
	c.member2 = uint32_t(__builtin_ctz(m)) & 3u;
	m = (m & (m - 1u)) | 0x10u;
	c.member3 = uint32_t(__builtin_ctz(m)) & 3u;
	return c;
}

// Packed word p of a class is word j of its rank-th block; NBLK words per block.
template <uint32_t NBLK>
__device__ __forceinline__ void class_word(const quad_class &c, uint32_t p, uint32_t &gp, uint32_t &j)
{
	const uint32_t rank = (p >= NBLK ? 1u : 0u) + (p >= 2 * NBLK ? 1u : 0u) + (p >= 3 * NBLK ? 1u : 0u);
	j = p - rank * NBLK;
	gp = rank == 0 ? c.member0 : rank == 1 ? c.member1 : rank == 2 ? c.member2 : c.member3;
}

// Lane p, p + 64, ... unpacks packed word p of the class.
template <typename E, int NB>
__device__ __forceinline__ void unpack_quad_words(const quad_class &c, const uint8_t *qb, uint32_t data_at,
						  int lane, typename E::C *slots, uint32_t first)
{
	using C = typename E::C;
	constexpr uint32_t NBLK = (BLOCK + 128 / NB - 1) / (128 / NB); // packed words per block
	const uint32_t P = c.count * NBLK;
	asm volatile("" : "+v"(lane)); // keep lane arithmetic of every width out of the callers' loops
#pragma unroll
	for (uint32_t r = 0; r < (QUAD * NBLK + WAVE - 1) / WAVE; r++) {
		const uint32_t p = uint32_t(lane) + r * WAVE;
		if (r * WAVE >= P)
			break;
		uint32_t gp, j;
		class_word<NBLK>(c, p, gp, j);
		// data_at of block gp lives in the lanes of group gp
		const uint32_t at = lane_gather(data_at, int(gp * GROUP));
		if (p < P) {
			const uint2 *q = reinterpret_cast<const uint2 *>(qb + at) + j * 2;
			const uint2 x = q[0], y = q[1];
			const uint32_t w[4] = {x.x, x.y, y.x, y.y};
			unpack_fields<C, NB>(w, slots + __umul24(gp - first, SLOT_STRIDE), int(j));
		}
	}
}

template <typename E, int LO, int HI>
__device__ __forceinline__ void unpack_quad_dispatch(uint32_t nb, const quad_class &c, const uint8_t *qb,
						     uint32_t data_at, int lane, typename E::C *slots,
						     uint32_t first)
{
	if constexpr (LO == HI) {
		unpack_quad_words<E, LO>(c, qb, data_at, lane, slots, first);
	} else {
		constexpr int MID = (LO + HI) / 2;
		if (nb <= uint32_t(MID))
			unpack_quad_dispatch<E, LO, MID>(nb, c, qb, data_at, lane, slots, first);
		else
			unpack_quad_dispatch<E, MID + 1, HI>(nb, c, qb, data_at, lane, slots, first);
	}
}

// One bit per group from a ballot of a group-uniform condition.
__device__ __forceinline__ uint32_t group_mask(uint64_t ballot)
{
	return (uint32_t(ballot) & 1u) | ((uint32_t(ballot) >> 15) & 2u) | ((uint32_t(ballot >> 32) & 1u) << 2) |
	       ((uint32_t(ballot >> 32) >> 13) & 8u);
}

// Decodes those blocks of the quad staged in qbuf (block g at byte rel[g], rel
// in lanes 0..4) that qualify for the four-block path and returns them as a
// bit mask; the caller does the others one by one.  The 256 values of block s
// of the quad go to sink(s, lane, v), lane l holding values 4l .. 4l+3.
// When all four blocks qualify, staged() is called once the quad's bytes are no
// longer needed (the last two blocks are unpacked but not yet handed out): the
// caller may overwrite qbuf there.
template <typename E, typename Sink, typename Staged>
__device__ __forceinline__ uint32_t decode_quad_fast(const uint32_t *qbuf, uint32_t qbytes, uint32_t rel,
						  int lane, typename E::C *slots, uint8_t *ends,
						  const uint4 *wt, Sink &&sink, Staged &&staged)
{
	using C = typename E::C;
	static_assert(sizeof(C) == 4, "element types up to 32 bits");
	const uint8_t *qb = reinterpret_cast<const uint8_t *>(qbuf);
	// Everything below that depends only on the lane number is the same for
	// every quad; left alone, the compiler computes it all once, for every
	// width, ahead of the quad loop and keeps it in registers (over 200 of
	// them).  An opaque lane number makes it recompute the little it needs.
	asm volatile("" : "+v"(lane));
	static_assert(GROUP == 16, "lane arithmetic below");
	const int g = int(uint32_t(lane) >> 4), sub = int(uint32_t(lane) & 15u);
	const uint32_t r0 = lane_gather(rel, g), r1 = lane_gather(rel, g + 1);
	const uint32_t len = r1 - r0;
	const bool inside = r0 + 16u <= qbytes && r1 <= qbytes && r0 < r1;
	const uint4 h = *reinterpret_cast<const uint4 *>(qb + (inside ? r0 : 0u));
	const quad_header m = parse_header<E>(h);
	const uint32_t nb = m.nb;
	const bool width_ok = nb >= uint32_t(FIELDS_MIN_NB) && nb <= uint32_t(E::W);
	const uint32_t pck = wt[width_ok ? nb : 1u].w; // bytes of the packed words of 256 values
	const uint32_t list_at = m.doff + pck;
	bool ok = inside && m.plain && width_ok && list_at <= len;
	// Exception lists, integer_codec.h:583-609: gap-coded positions, then high
	// parts, each either bit-packed or as varints.  One exception per lane
	// and round; varints are located through a 64-byte window, which holds
	// at least 16 of them.
	const bool is_pfr = m.enc == OROCH_BITPFR;
	const uint4 iw = wt[m.inb && m.inb <= 32u ? m.inb : 1u], vw = wt[m.vnb && m.vnb <= 32u ? m.vnb : 1u];
	// bitpck.h:61-71 with the 16-bit reciprocal (exact below 512; a count beyond
	// QUAD_MAX_EXC fails the test further down whatever this gives)
	const uint32_t ilen_pck = (uint32_t(__umul24(m.nout + iw.x - 1u, iw.z)) >> 16) << 4;
	const uint32_t scan_rel = list_at + (m.inb ? ilen_pck : 0u);        // where varints start
	if (is_pfr) {
		// one exception per lane of the group; a block with more goes the
		// one-block way, where the whole wavefront shares the lists
		ok = ok && m.nout >= 1 && m.nout <= QUAD_MAX_EXC && m.inb <= 32u && m.vnb <= uint32_t(E::W);
		if (!m.vnb)
			ok = ok && scan_rel <= len && len - scan_rel <= uint32_t(LIST_BYTES);
		else
			ok = ok && scan_rel + 16u <= len;
	}
	const uint32_t fast = group_mask(__ballot(ok));
	// (a single block that qualifies is cheaper one block at a time than through
	// this path, whose cost hardly depends on how many blocks it carries)
	if ((fast & (fast - 1u)) == 0)
		return 0;

	// The exceptions of all four blocks first: position and shifted high part
	// of exception `sub` of the group's block, to be OR-ed into the value slots.
	const bool has_exc = ok && is_pfr;
	bool hit = false;
	uint32_t pos = 0;
	C patch = 0;
	if (__ballot(has_exc) != 0) {
		const uint32_t lb = r0 + list_at;      // index list; a multiple of 8
		const uint32_t sb = r0 + scan_rel;     // first varint, if any list has them
		uint8_t *my_ends = ends + __umul24(uint32_t(g), ENDS);
		if (__ballot(has_exc && (!m.inb || !m.vnb)) != 0) {
			// every lane looks at four bytes of the window and files the end
			// position of each varint that ends there under the varint's number
			const uint32_t w = qbuf[(sb >> 2) + uint32_t(sub)];
			const uint32_t t = ~w & 0x80808080u;
			const uint32_t cnt = uint32_t(__popc(t));
			const uint32_t before = row_scan_sum(cnt) - cnt;
#pragma unroll
			for (int b = 0; b < 4; b++) {
				const bool term = (t >> (8 * b + 7)) & 1u;
				const uint32_t rank = before + uint32_t(__popc(t & ((1u << (8 * b + 7)) - 1u)));
				const uint32_t slot = term && rank < 2 * QUAD_MAX_EXC ? rank + 1u : uint32_t(ENDS - 4 + b);
				my_ends[slot] = uint8_t(4 * sub + b + 1);
			}
			wave_sync();
		}
		// where the value list starts when the index list is varints: after
		// varint number nout - 1
		const uint32_t vb = m.inb ? sb : sb + (has_exc ? my_ends[m.nout] : 0u);
		const uint32_t voff = m.inb ? 0u : m.nout; // number of the first value varint
		const uint32_t e = uint32_t(sub);
		const bool active = has_exc && e < m.nout;
		uint32_t gap = 0, hi = 0;
		if (__ballot(active && m.inb) != 0) {
			const uint32_t bk = uint32_t(__umul24(e, iw.z)) >> 16;
			gap = lds_get_bits32(qbuf, lb * 8u + bk * 128u + (e - bk * iw.x) * m.inb, m.inb ? m.inb : 1u);
		}
		if (__ballot(active && !m.inb) != 0) {
			const uint32_t is = active && e && !m.inb ? my_ends[e] : 0u;
			const uint32_t x = varint_at(qbuf, sb + is);
			gap = m.inb ? gap : x;
		}
		if (__ballot(active && m.vnb) != 0) {
			const uint32_t bk = uint32_t(__umul24(e, vw.z)) >> 16;
			hi = lds_get_bits32(qbuf, vb * 8u + bk * 128u + (e - bk * vw.x) * m.vnb, m.vnb ? m.vnb : 1u);
		}
		if (__ballot(active && !m.vnb) != 0) {
			const uint32_t vs = active && !m.vnb && voff + e ? my_ends[voff + e] : 0u;
			const uint32_t x = varint_at(qbuf, sb + vs);
			hi = m.vnb ? hi : x;
		}
		// positions: running sum of gap + 1 (offset.h:62-67)
		pos = row_scan_sum(active ? gap + 1u : 0u) - 1u;
		hit = active && pos < uint32_t(BLOCK);
		patch = C(hi << (nb & 31u)) & E::MASK;
	}

	// Two blocks at a time through the value slots (half the LDS of four, which
	// buys a wavefront more per SIMD): unpack, patch, hand out.
#pragma unroll
	for (int half = 0; half < 2; half++) {
		const uint32_t mine = fast & (3u << (2 * half));
		if (mine == 0)
			continue;
		const bool in_half = ok && (g >> 1) == half;
		// blocks of one width are unpacked together
		for (uint32_t left = mine; left;) {
			const uint32_t nb_u = read_lane(nb, int(__builtin_ctz(left)) * GROUP);
			const uint32_t cls = group_mask(__ballot(in_half && nb == nb_u));
			unpack_quad_dispatch<E, FIELDS_MIN_NB, E::W>(nb_u, make_class(cls), qb, r0 + m.doff, lane, slots,
								     uint32_t(2 * half));
			left &= ~cls;
		}
		wave_sync();
		// lanes with nothing to patch here OR zero into a spare slot
		const bool now = hit && in_half;
		atomicOr(&slots[now ? __umul24(uint32_t(g & 1), SLOT_STRIDE) + pos : uint32_t(2 * SLOT_STRIDE - 1)],
			 now ? patch : C(0));
		wave_sync();
		if (half == 1 && fast == 15u)
			staged();
		// every lane picks up four values of each block; origin and encoding
		// of block s come from the first lane of its group
#pragma unroll
		for (int s = 2 * half; s < 2 * half + 2; s++) {
			if (!((fast >> s) & 1u))
				continue;
			const uint32_t origin = read_lane(m.origin, s * GROUP);
			const uint32_t enc = read_lane(m.enc, s * GROUP);
			const uint4 q = *reinterpret_cast<const uint4 *>(&slots[(s & 1) * SLOT_STRIDE + lane * VPL]);
			C v[VPL] = {q.x, q.y, q.z, q.w};
#pragma unroll
			for (int i = 0; i < VPL; i++)
				v[i] = enc == OROCH_BITPCK ? unzz<E>(v[i] & E::MASK) : C((v[i] + origin) & E::MASK);
			sink(s, lane, v);
		}
		wave_sync();
	}
	return fast;
}


// varint.h:140-147 for the varint and varfor blocks of a quad (integer_codec.h
// :628-640: the zigzag or the origin value codec), which the path above leaves
// out: two blocks at a time, 32 lanes to a block.  A lane takes a run of the
// block's bytes, counts the values that end in it (a byte with its top bit
// clear ends one), learns from a scan which value its run starts in the middle
// of, and walks its bytes; the run before it ends at most four bytes into a
// value, which the lane picks up from there.  The values go through the slots
// to sink() as above.  skip: blocks already decoded.  Returns the blocks done.
template <typename E, typename Sink>
__device__ __forceinline__ uint32_t decode_quad_varints(const uint32_t *qbuf, uint32_t qbytes, uint32_t rel,
							     uint32_t skip, int lane, typename E::C *slots, Sink &&sink)
{
	using C = typename E::C;
	static_assert(sizeof(C) == 4, "element types up to 32 bits");
	const uint8_t *qb = reinterpret_cast<const uint8_t *>(qbuf);
	asm volatile("" : "+v"(lane)); // see decode_quad_fast
	const int g = int(uint32_t(lane) >> 4);
	const uint32_t r0 = lane_gather(rel, g), r1 = lane_gather(rel, g + 1);
	const uint32_t len = r1 - r0;
	const bool inside = r0 + 16u <= qbytes && r1 <= qbytes && r0 < r1;
	// [enc] or [enc][origin varint <= 5 bytes], then zeros up to byte 8 (integer_group.h:70-72)
	const uint2 h = *reinterpret_cast<const uint2 *>(qb + (inside ? r0 : 0u));
	const uint32_t enc = h.x & 0xffu;
	// (a value takes a byte at least)
	const bool vok = inside && (enc == OROCH_VARINT || enc == OROCH_VARFOR) && len >= 8u + uint32_t(BLOCK) &&
			 !((skip >> g) & 1u);
	const uint32_t todo = group_mask(__ballot(vok));
	if (todo == 0)
		return 0;
	bool more;
	uint32_t olen;
	uint32_t zo = varint4(__builtin_amdgcn_alignbyte(h.y, h.x, 1), &more, &olen);
	if (more)
		zo |= ((h.y >> 8) & 0x0fu) << 28;
	const uint32_t origin = unzz<E>(zo & E::MASK);

	const bool upper = (uint32_t(lane) >> 5) != 0;
	const uint32_t l32 = uint32_t(lane) & 31u;
#pragma unroll
	for (int half = 0; half < 2; half++) {
		if ((todo & (3u << (2 * half))) == 0)
			continue;
		// what the lanes of my block's group know about it
		auto of_block = [&](uint32_t x) {
			const uint32_t a = read_lane(x, 2 * half * GROUP), b = read_lane(x, (2 * half + 1) * GROUP);
			return upper ? b : a;
		};
		const bool mine = of_block(uint32_t(vok)) != 0;
		const uint32_t at0 = of_block(r0) + 8u, blen = of_block(len) - 8u;
		// bytes per lane: a multiple of four; 32 of them cover the block's image
		const uint32_t run = mine ? (((blen + 31u) >> 5) + 3u) & ~3u : 0u;
		const uint32_t first = at0 + __umul24(l32, run), stop = at0 + blen;
		const uint32_t nd = wave_max(run >> 2);
		uint32_t ends_here = 0;
		for (uint32_t k = 0; k < nd; k++) {
			const uint32_t o = first + 4u * k;
			const bool live = 4u * k < run && o < stop;
			const uint32_t w = live ? qbuf[o >> 2] : 0x80808080u;
			ends_here += uint32_t(__popc(~w & 0x80808080u));
		}
		// values that end before my run, over the 32 lanes of the block
		uint32_t before = row_scan_sum(ends_here);
		before += dpp_mov<0x142, 0xa>(0u, before);
		uint32_t rank = before - ends_here;
		// the bytes of the value under way where my run starts: what follows the
		// last ending byte among the four bytes before it
		uint32_t acc = 0, shift = 0;
		if (mine && l32 != 0 && first < stop) {
			const uint32_t p = qbuf[(first >> 2) - 1u];
			const uint32_t t = ~p & 0x80808080u;
			const uint32_t keep = t ? uint32_t(__clz(t)) >> 3 : 4u;
			const uint32_t tailb = keep ? p >> (8u * (4u - keep)) : 0u;
			acc = (tailb & 0x7fu) | ((tailb >> 1) & 0x3f80u) | ((tailb >> 2) & 0x1fc000u) |
			      ((tailb >> 3) & 0xfe00000u);
			shift = 7u * keep;
		}
		C *vals = slots + (upper ? SLOT_STRIDE : 0);
		for (uint32_t k = 0; k < nd; k++) {
			const uint32_t o = first + 4u * k;
			const bool live = mine && 4u * k < run && o < stop;
			const uint32_t w = live ? qbuf[o >> 2] : 0x80808080u;
#pragma unroll
			for (int b = 0; b < 4; b++) {
				const uint32_t byte = (w >> (8 * b)) & 0xffu;
				acc |= shift < 32u ? (byte & 0x7fu) << shift : 0u;
				shift += 7u;
				if (live && !(byte & 0x80u)) {
					if (rank < uint32_t(BLOCK))
						vals[rank] = C(acc);
					rank++;
					acc = 0;
					shift = 0;
				}
			}
		}
		wave_sync();
#pragma unroll
		for (int s = 2 * half; s < 2 * half + 2; s++) {
			if (!((todo >> s) & 1u))
				continue;
			const uint32_t org = read_lane(origin, s * GROUP);
			const uint32_t e = read_lane(enc, s * GROUP);
			const uint4 q = *reinterpret_cast<const uint4 *>(&slots[(s & 1) * SLOT_STRIDE + lane * VPL]);
			C v[VPL] = {q.x, q.y, q.z, q.w};
#pragma unroll
			for (int i = 0; i < VPL; i++)
				v[i] = e == OROCH_VARINT ? unzz<E>(v[i] & E::MASK) : C((v[i] + org) & E::MASK);
			sink(s, lane, v);
		}
		wave_sync();
	}
	return todo;
}

// Persistent wavefronts over quads of blocks; offsets are fetched two quads
// ahead and the quad's bytes one quad ahead (the first 3 KiB of them, into
// registers), so the dependent latencies overlap with decoding.  The loop is
// latency-bound (a chain of LDS round trips per quad): registers and LDS are
// kept to what five wavefronts per SIMD allow.
//
// Where the loads are waited for matters.  The compiler's wait counts assume
// the fewest younger memory operations any path can have issued, and the
// stores of a quad sit under conditions, so a wait for a load that has stores
// after it is a wait for everything, stores included.  The next quad's bytes
// are therefore awaited and moved to LDS as soon as this quad's are no longer
// needed, ahead of its last stores, and the loads for the quad after that are
// issued there too; what is younger than those loads at their wait are stores
// that have had most of a quad's decoding to complete.  All loads of the walk
// are unconditional (clamped addresses) for the same reason.
//
// want(first block of the quad) is asked before a quad is decoded: WALK_STOP
// ends this wavefront's walk.  The values of block blk (nv of them) go to
// sink(blk, nv, lane, v), lane l holding values 4l .. 4l+3.
template <typename E, typename Want, typename Sink>
__device__ __forceinline__ void walk_quads(const uint8_t *__restrict__ src,
					    const uint64_t *__restrict__ offsets, size_t n, size_t nblocks,
					    uint64_t stream_limit, Want want, Sink sink)
{
	using C = typename E::C;
	constexpr uint32_t IMG = uint32_t(image_bytes<E>() - 16); // longest block image
	constexpr uint32_t QBYTES = QUAD * IMG;
	constexpr int NCH = int((QBYTES + 1023) / 1024);
	constexpr int NPRE = NCH < 3 ? NCH : 3;
	__shared__ __attribute__((aligned(16))) uint32_t qbufs[WAVES_PER_WG][QBYTES / 4 + 24];
	// value slots of two blocks; the one-block path takes one block of slots and
	// one of list positions
	constexpr int SLOTS = 2 * SLOT_STRIDE;
	static_assert(SLOTS >= SLOT_STRIDE + BLOCK, "slots and positions of one block");
	__shared__ __attribute__((aligned(16))) C slotss[WAVES_PER_WG][SLOTS];
	__shared__ __attribute__((aligned(16))) uint8_t endss[WAVES_PER_WG][QUAD * ENDS];
	__shared__ uint4 wt[65];
	const int wave = uniform(int(threadIdx.x / WAVE)), lane = lane_id();
	load_width_table(wt);
	uint32_t *qbuf = qbufs[wave];
	uint8_t *qb = reinterpret_cast<uint8_t *>(qbuf);
	C *slots = slotss[wave];
	const size_t nquads = (nblocks + QUAD - 1) / QUAD;
	const size_t stride = size_t(gridDim.x) * WAVES_PER_WG;

	// Offsets come from the caller's table and may be anything: the quad is
	// confined to [0, offsets[nblocks]), the length of the stream, and to the
	// caller's stream_limit, so a damaged table yields wrong values but no read
	// outside the stream.
	const uint64_t limit = min_u64(uniform(offsets[nblocks]), stream_limit);
	// a stream without a single 16-byte piece: the loads below go to the table
	// instead (it has two entries at least) and every quad has length zero
	const uint8_t *bytes = limit >= 16 ? src : reinterpret_cast<const uint8_t *>(offsets);
	const uint64_t last_piece = limit >= 16 ? (limit - 16) & ~uint64_t(15) : 0;

	// lanes 0..4 hold the offsets of the quad's blocks and its end (the end of
	// the stream for quads past the last)
	auto extent = [&](size_t q, uint64_t &o) {
		size_t i = (q < nquads ? q * QUAD : nblocks) + size_t(lane < QUAD ? lane : QUAD);
		i = i < nblocks ? i : nblocks;
		o = __builtin_nontemporal_load(offsets + i);
	};
	auto resolve = [&](uint64_t o, uint64_t &start, uint32_t &rel, uint32_t &total) {
		start = min_u64(read_lane(o, 0), limit) & ~uint64_t(15);
		const uint64_t room = limit - start;
		uint64_t d = o >= start ? o - start : 0; // per lane
		d = d < room ? d : room;
		rel = clamp_u64(d, QBYTES) & ~15u; // images start at multiples of 16
		total = read_lane(rel, QUAD);
	};
	// The bytes in flight, one 1 KiB piece per variable (named variables: as an
	// array they end up in scratch memory once the register budget is tight).
	// Pieces past the quad's length are loaded from its first piece instead
	// and dropped when the quad is staged; no piece reaches past the stream.
	static_assert(NPRE <= 3, "pieces in flight");
	uint4 im0, im1, im2;
	auto piece = [&](uint64_t start, uint32_t total, int c) {
		const uint32_t o = uint32_t(c) * 1024u + uint32_t(lane) * 16u;
		const uint64_t at = start + (o < total ? o : 0u);
		return *reinterpret_cast<const uint4 *>(bytes + (at < last_piece ? at : last_piece));
	};
	auto fetch = [&](uint64_t start, uint32_t total) {
		im0 = piece(start, total, 0);
		if constexpr (NPRE > 1)
			im1 = piece(start, total, 1);
		if constexpr (NPRE > 2)
			im2 = piece(start, total, 2);
	};
	auto put = [&](uint32_t total, int c, const uint4 &x) {
		const uint32_t o = uint32_t(c) * 1024u + uint32_t(lane) * 16u;
		if (o < total)
			*reinterpret_cast<uint4 *>(qb + o) = x;
	};
	auto stage = [&](uint64_t start, uint32_t total) {
		put(total, 0, im0);
		if constexpr (NPRE > 1)
			put(total, 1, im1);
		if constexpr (NPRE > 2)
			put(total, 2, im2);
		// a quad longer than what was fetched ahead: the rest, now
		for (uint32_t o = uint32_t(NPRE) * 1024u + uint32_t(lane) * 16u; o < total; o += 1024u)
			*reinterpret_cast<uint4 *>(qb + o) = *reinterpret_cast<const uint4 *>(src + start + o);
	};

	size_t q = size_t(blockIdx.x) * WAVES_PER_WG + wave;
	// rel0/total0: the quad staged in LDS; start1/rel1/total1 and im: the next
	// quad, in flight; o1: the offsets of the one after it, in flight
	uint64_t o0, o1, start0, start1;
	uint32_t rel0, rel1, total0, total1;
	extent(q, o0);
	extent(q + stride, o1);
	resolve(o0, start0, rel0, total0);
	fetch(start0, total0);
	stage(start0, total0);
	wave_sync();
	resolve(o1, start1, rel1, total1);
	fetch(start1, total1);
	extent(q + 2 * stride, o1);
	// moves the next quad in and sets the loads for the one after it going
	auto advance = [&]() {
		stage(start1, total1);
		rel0 = rel1;
		total0 = total1;
		resolve(o1, start1, rel1, total1);
		fetch(start1, total1);
		extent(q + 3 * stride, o1);
	};
	for (; q < nquads; q += stride) {
		const size_t blk0 = q * QUAD;
		const size_t base = blk0 * BLOCK;
		if (want(blk0) == WALK_STOP)
			break;
		uint32_t done = 0;
		if (blk0 + QUAD <= nblocks && base + size_t(QUAD) * BLOCK <= n)
			done = decode_quad_fast<E>(
				qbuf, total0, rel0, lane, slots, endss[wave], wt,
				[&](int s, int ln, const C(&v)[VPL]) { sink(blk0 + size_t(s), BLOCK, ln, v); }, advance);
		if (done != 15u) {
			// varint blocks together, the others one by one, from the same staged bytes
			if (blk0 + QUAD <= nblocks && base + size_t(QUAD) * BLOCK <= n)
				done |= decode_quad_varints<E>(
					qbuf, total0, rel0, done, lane, slots,
					[&](int s, int ln, const C(&v)[VPL]) { sink(blk0 + size_t(s), BLOCK, ln, v); });
			for (int b = 0; b < QUAD && blk0 + size_t(b) < nblocks; b++) {
				if ((done >> b) & 1u)
					continue;
				const uint32_t at = read_lane(rel0, b), end = read_lane(rel0, b + 1);
				const uint32_t len = end > at ? end - at : 0u;
				const int nv = block_count(n, blk0 + size_t(b));
				C v[VPL];
				uint32_t *img = qbuf + (at >> 2);
				C *val = slots;
				uint32_t *gp = reinterpret_cast<uint32_t *>(slots + SLOT_STRIDE);
				if (nv == BLOCK)
					decode_block<E, true>(img, len, nv, lane, val, gp, wt, v);
				else
					decode_block<E, false>(img, len, nv, lane, val, gp, wt, v);
				sink(blk0 + size_t(b), nv, lane, v);
				wave_sync();
			}
			advance();
		}
		wave_sync();
	}
}

template <typename E>
__global__ __launch_bounds__(WAVE *WAVES_PER_WG) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_decode_quad(
	const uint8_t *__restrict__ src, const uint64_t *__restrict__ offsets, size_t n, size_t nblocks,
	uint64_t stream_limit, typename E::S *__restrict__ dst)
{
	using C = typename E::C;
	walk_quads<E>(
		src, offsets, n, nblocks, stream_limit, [](size_t) { return WALK_DECODE; },
		[&](size_t blk, int nv, int lane, const C(&v)[VPL]) { store_block<E>(dst, blk * BLOCK, nv, lane, v); });
}

} // namespace oroch_hip
