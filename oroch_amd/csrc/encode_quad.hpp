// encode_quad.hpp -- the encoder's fast path: four consecutive blocks per
// wavefront, sixteen lanes per block (the counterpart of decode_quad.hpp).
//
// Works from the descriptors the selection pass left behind.  Lane (g, sub)
// holds values 16 sub .. 16 sub + 15 of block g; the deltas go through an LDS
// slot array, from which lane p of the wavefront packs 16-byte word p of the
// quad with compile-time field positions and stores it straight to the stream.
// The header is produced one byte per lane, exception lists are written by
// the lanes that own exceptions.  Covers bitpck, bitfor and bitpfr blocks from
// four bits up; varint and varfor blocks of the quad follow in a pass of their
// own; anything else goes block by block through encode_block.
//
// Included by group.hip after encode_block and decode_quad.hpp; 8-, 16- and
// 32-bit element types.
#pragma once

namespace oroch_hip {

constexpr int VPQ = BLOCK / GROUP;   // values per lane and block
constexpr int TAIL_BYTES = 256;      // exception lists + padding a group stages in LDS

// Inclusive prefix maximum within each row of 16 lanes; values are >= -1.
__device__ __forceinline__ int row_scan_max(int x)
{
	int y;
	y = dpp_mov<0x111>(-1, x); x = x > y ? x : y;
	y = dpp_mov<0x112>(-1, x); x = x > y ? x : y;
	y = dpp_mov<0x114>(-1, x); x = x > y ? x : y;
	y = dpp_mov<0x118>(-1, x); x = x > y ? x : y;
	return x;
}

// The same for unsigned values; with 0 as the identity each step is one
// max instruction that takes its operand through the lane shift.
__device__ __forceinline__ uint32_t row_scan_max(uint32_t x)
{
	uint32_t y;
	y = dpp_mov<0x111>(0u, x); x = x > y ? x : y;
	y = dpp_mov<0x112>(0u, x); x = x > y ? x : y;
	y = dpp_mov<0x114>(0u, x); x = x > y ? x : y;
	y = dpp_mov<0x118>(0u, x); x = x > y ? x : y;
	return x;
}

// Lane p, p + 64, ... packs word p of a class of blocks (see class_word) from
// the slots and stores it at its place in the stream.
template <typename E, int NB>
__device__ __forceinline__ void pack_quad_words(const quad_class &c, const typename E::C *slots,
						uint32_t data_at, int lane, uint8_t *out)
{
	using C = typename E::C;
	constexpr uint32_t NBLK = (BLOCK + 128 / NB - 1) / (128 / NB);
	const uint32_t P = c.count * NBLK;
	asm volatile("" : "+v"(lane)); // keep lane arithmetic of every width out of the callers' loops
#pragma unroll
	for (uint32_t r = 0; r < (QUAD * NBLK + WAVE - 1) / WAVE; r++) {
		const uint32_t p = uint32_t(lane) + r * WAVE;
		if (r * WAVE >= P)
			break;
		uint32_t gp, j;
		class_word<NBLK>(c, p, gp, j);
		const uint32_t at = lane_gather(data_at, int(gp * GROUP));
		if (p < P) {
			uint32_t w[4];
			pack_fields<C, NB, true>(slots + __umul24(gp, SLOT_STRIDE), int(j), w);
			*reinterpret_cast<uint4 *>(out + at + j * 16u) = make_uint4(w[0], w[1], w[2], w[3]);
		}
	}
}

template <typename E, int LO, int HI>
__device__ __forceinline__ void pack_quad_dispatch(uint32_t nb, const quad_class &c,
						   const typename E::C *slots, uint32_t data_at, int lane,
						   uint8_t *out)
{
	if constexpr (LO == HI) {
		pack_quad_words<E, LO>(c, slots, data_at, lane, out);
	} else {
		constexpr int MID = (LO + HI) / 2;
		if (nb <= uint32_t(MID))
			pack_quad_dispatch<E, LO, MID>(nb, c, slots, data_at, lane, out);
		else
			pack_quad_dispatch<E, MID + 1, HI>(nb, c, slots, data_at, lane, out);
	}
}

// da, db: the two halves of this lane's block descriptor (put_desc layout);
// off: the block's stream offset.  Encodes the blocks of the quad that qualify
// and returns them as a bit mask; the caller does the others one by one.
template <typename E>
__device__ __forceinline__ uint32_t encode_quad_fast(const typename E::C (&v)[VPQ], const uint4 da,
						  const uint4 db, uint64_t off, int lane,
						  typename E::C *slots, uint8_t *tails, uint8_t *heads,
						  const uint4 *wt, uint8_t *dst)
{
	using C = typename E::C;
	static_assert(sizeof(C) == 4, "element types up to 32 bits");
	asm volatile("" : "+v"(lane)); // see decode_quad_fast
	static_assert(GROUP == 16, "lane arithmetic below");
	const int g = int(uint32_t(lane) >> 4), sub = int(uint32_t(lane) & 15u);

	const C origin = da.x;
	const uint32_t dataspace = da.z, size = da.w;
	const uint32_t metaspace = db.x & 0xffffu, nout = db.x >> 16;
	const uint32_t enc = (db.y >> 16) & 0xffu, nb = db.y >> 24;
	const uint32_t ienc = db.z & 0xffu, inb = (db.z >> 8) & 0xffu;
	const uint32_t venc = (db.z >> 16) & 0xffu, vnb = db.z >> 24;
	const uint32_t doff = (metaspace + 7u) & ~7u; // integer_group.h:50-52
	const bool width_ok = nb >= uint32_t(FIELDS_MIN_NB) && nb <= uint32_t(E::W);
	const uint32_t pck = wt[width_ok ? nb : 1u].w;
	const uint32_t list_at = doff + pck;
	bool ok = (enc == OROCH_BITPCK || enc == OROCH_BITFOR || enc == OROCH_BITPFR) && width_ok &&
		  dataspace >= pck && size >= list_at && size - list_at <= uint32_t(TAIL_BYTES);
	const uint32_t fast = group_mask(__ballot(ok));
	// (a single block that qualifies is cheaper one block at a time than through
	// this path, whose cost hardly depends on how many blocks it carries)
	if ((fast & (fast - 1u)) == 0)
		return 0;

	// stream positions relative to the quad's first block
	const uint64_t start = uniform(off);
	const uint32_t rel = uint32_t(off - start);
	uint8_t *out = dst + start;

	// deltas (or zigzag images) of all values into the slots, in full: the
	// packer cuts them to the width, the exception lists need the rest
	C *mine = slots + __umul24(uint32_t(g), SLOT_STRIDE); // (full-rate multiply)
	const bool has_exc = ok && enc == OROCH_BITPFR;
	const bool zig = E::sgn && enc == OROCH_BITPCK;
	const C org = enc == OROCH_BITPCK ? C(0) : origin;
	C u[VPQ];
	if (E::sgn && __ballot(zig) != 0) {
#pragma unroll
		for (int i = 0; i < VPQ; i++)
			u[i] = zig ? zz<E>(v[i]) : C((v[i] - org) & E::MASK);
	} else {
#pragma unroll
		for (int i = 0; i < VPQ; i++)
			u[i] = C((v[i] - org) & E::MASK);
	}
#pragma unroll
	for (int k = 0; k < VPQ / 4; k++)
		*reinterpret_cast<uint4 *>(&mine[sub * VPQ + 4 * k]) = make_uint4(u[4 * k], u[4 * k + 1], u[4 * k + 2], u[4 * k + 3]);
	// the last packed word of a block reads past value 255: zeros there
	*reinterpret_cast<uint2 *>(&mine[BLOCK + 2 * sub]) = make_uint2(0, 0);
	// exception list staging starts out as zeros (it also supplies the padding)
	uint8_t *tail = tails + g * TAIL_BYTES;
	*reinterpret_cast<uint4 *>(tail + 16 * sub) = make_uint4(0, 0, 0, 0);
	// which of my values are exceptions
	uint32_t m = 0;
	if (__ballot(has_exc) != 0) {
		// u >= 2^nb exactly when u + (2^32 - 2^nb) carries; the carries are
		// shifted into m from the top value down (an add with carry-out and
		// an add with carry-in per value)
		const uint32_t lift = 0u - (1u << (nb & 31u));
#pragma unroll
		for (int i = VPQ - 1; i >= 0; i--) {
			uint32_t sum;
			const uint32_t carry = __builtin_add_overflow(uint32_t(u[i]), lift, &sum) ? 1u : 0u;
			m = m + m + carry;
		}
		m = has_exc && nb < 32u ? m : 0u;
	}
	wave_sync();

	// blocks of one width are packed together
	for (uint32_t left = fast; left;) {
		const uint32_t nb_u = read_lane(nb, int(__builtin_ctz(left)) * GROUP);
		const uint32_t cls = group_mask(__ballot(ok && nb == nb_u));
		pack_quad_dispatch<E, FIELDS_MIN_NB, E::W>(nb_u, make_class(cls), slots, rel + doff, lane, out);
		left &= ~cls;
	}

	// metadata, integer_codec.h:128-151, 177-182, 196-204: lane `sub` makes
	// byte `sub` of the (at most 11-byte) header
	{
		const bool has_o = enc != OROCH_BITPCK;
		const C zo = zz<E>(origin);
		uint32_t olen = (bitlen(zo) + 6u) / 7u;
		olen = has_o ? (olen ? olen : 1u) : 0u;
		const uint32_t pos = uint32_t(sub);
		const uint32_t p_nb = 1u + olen, p_nout = p_nb + 1u;
		const uint32_t nlen = nout >= 128u ? 2u : 1u;
		const uint32_t p_i = p_nout + nlen, p_v = p_i + 1u;
		uint32_t byte = 0;
		if (pos == 0) {
			byte = enc;
		} else if (pos <= olen) {
			const uint32_t k = pos - 1u;
			byte = (uint32_t(zo >> (7u * k)) & 0x7fu) | (k + 1u < olen ? 0x80u : 0u);
		} else if (pos == p_nb) {
			byte = nb;
		} else if (has_exc) {
			if (pos == p_nout)
				byte = (nout & 0x7fu) | (nlen == 2u ? 0x80u : 0u);
			else if (nlen == 2u && pos == p_nout + 1u)
				byte = nout >> 7;
			else if (pos == p_i)
				byte = ienc == OROCH_BITPCK ? inb : 0u;
			else if (pos == p_v)
				byte = venc == OROCH_BITPCK ? vnb : 0u;
		}
		uint8_t *head = heads + g * 16;
		head[sub] = uint8_t(byte);
		wave_sync();
		if (ok && uint32_t(sub) * 8u < doff)
			*reinterpret_cast<uint2 *>(out + rel + sub * 8) = *reinterpret_cast<const uint2 *>(head + sub * 8);
	}

	// exception lists, integer_codec.h:563-581: gap-coded positions
	// (offset.h:55-60) then high parts, bit-packed or as varints
	if (__ballot(has_exc) != 0) {
		const uint32_t cnt = uint32_t(__popc(m));
		const uint32_t before = row_scan_sum(cnt) - cnt;
		const int last = m ? sub * VPQ + 31 - int(__clz(m)) : -1;
		const int upto = row_scan_max(last);
		const int prev0 = dpp_mov<0x111>(-1, upto); // last exception in lower lanes
		const uint4 iw = wt[inb ? inb : 1u], vw = wt[vnb ? vnb : 1u];
		// first pass over my exceptions: how many list bytes they take.  A
		// gap takes two varint bytes from 128 on, which only the gap in front
		// of a lane's first exception can reach.
		const uint32_t gfirst = uint32_t(sub * VPQ + int(__builtin_ctz(m | 0x10000u)) - prev0 - 1);
		const uint32_t ib = cnt + (m && gfirst >= 128u ? 1u : 0u);
		uint32_t vb = 0;
		if (__ballot(has_exc && venc != OROCH_BITPCK) != 0) {
			uint32_t mm = m;
			while (__ballot(mm != 0) != 0) {
				if (mm) {
					const int pos = sub * VPQ + int(__builtin_ctz(mm));
					vb += vsp(C(mine[pos] >> nb));
					mm &= mm - 1u;
				}
			}
		}
		const uint32_t isum = row_scan_sum(ib), vsum = row_scan_sum(vb);
		// (bitpck.h:61-71 with the 16-bit reciprocal: nout + capacity - 1 < 512)
		const uint32_t ilen = ienc == OROCH_BITPCK ? (uint32_t(__umul24(nout + iw.x - 1u, iw.z)) >> 16) << 4
							    : lane_gather(isum, g * GROUP + GROUP - 1);
		uint32_t io = isum - ib, vo = ilen + vsum - vb, r = before;
		uint32_t *tail32 = reinterpret_cast<uint32_t *>(tail);
		{
			uint32_t mm = m;
			int prev = prev0;
			while (__ballot(mm != 0) != 0) {
				if (mm) {
					const int pos = sub * VPQ + int(__builtin_ctz(mm));
					const uint32_t gap = uint32_t(pos - prev - 1);
					prev = pos;
					const C hi = mine[pos] >> nb;
					if (ienc == OROCH_BITPCK) {
						const uint32_t bk = uint32_t(__umul24(r, iw.z)) >> 16;
						lds_or_bits(tail32, bk * 128u + (r - bk * iw.x) * inb, gap);
					} else {
						tail[io++] = uint8_t((gap & 0x7fu) | (gap >= 128u ? 0x80u : 0u));
						if (gap >= 128u)
							tail[io++] = uint8_t(gap >> 7);
					}
					if (venc == OROCH_BITPCK) {
						const uint32_t bk = uint32_t(__umul24(r, vw.z)) >> 16;
						lds_or_bits(tail32, ilen * 8u + bk * 128u + (r - bk * vw.x) * vnb, hi);
					} else {
						vo = put_varint(tail, vo, uint64_t(hi));
					}
					r++;
					mm &= mm - 1u;
				}
			}
		}
	}
	wave_sync();
	// lists and the zero padding up to the block's 16-byte boundary
	if (ok) {
		for (uint32_t k = uint32_t(sub) * 8u; k < size - list_at; k += GROUP * 8u)
			*reinterpret_cast<uint2 *>(out + rel + list_at + k) = *reinterpret_cast<const uint2 *>(tail + k);
	}
	return fast;
}

// The varint and varfor blocks of a quad (varint.h:133-139 with the zigzag or
// the origin value codec, integer_codec.h:511-518), which the path above
// leaves out.  Every lane lays the varints of its sixteen values end to end
// into the block's slots, which serve as the staging area of its bytes, and
// the area goes to the stream with the zero padding up to the block's 16-byte
// boundary.  Loads its own operands: it runs off the main path.  Returns the
// blocks done.
template <typename E>
__device__ __forceinline__ uint32_t encode_quad_varints(const typename E::S *__restrict__ src,
							     const oroch_block_desc *__restrict__ desc,
							     const uint64_t *__restrict__ offsets, size_t blk0, int lane,
							     typename E::C *slots, uint8_t *__restrict__ dst)
{
	using C = typename E::C;
	static_assert(sizeof(C) == 4, "element types up to 32 bits");
	asm volatile("" : "+v"(lane)); // see decode_quad_fast
	const int g = int(uint32_t(lane) >> 4), sub = int(uint32_t(lane) & 15u);
	const size_t blk = blk0 + size_t(g);
	const uint4 *dq = reinterpret_cast<const uint4 *>(desc + blk);
	const uint4 da = dq[0], db = dq[1];
	const C origin = da.x;
	const uint32_t dataspace = da.z, size = da.w;
	const uint32_t metaspace = db.x & 0xffffu;
	const uint32_t enc = (db.y >> 16) & 0xffu;
	const uint32_t doff = (metaspace + 7u) & ~7u; // integer_group.h:50-52
	const bool vok = (enc == OROCH_VARINT || enc == OROCH_VARFOR) && doff == 8u &&
			 dataspace + 16u <= uint32_t(sizeof(C)) * SLOT_STRIDE && size >= doff + dataspace;
	const uint32_t todo = group_mask(__ballot(vok));
	if (todo == 0)
		return 0;
	uint8_t *out = dst + offsets[blk];
	C u[VPQ];
	load16<E>(src + blk * BLOCK + size_t(sub) * VPQ, u);
	uint32_t *stage = reinterpret_cast<uint32_t *>(slots + __umul24(uint32_t(g), SLOT_STRIDE));
	// zeros: the varints are OR-ed in, and what follows them is padding
	static_assert(SLOT_STRIDE * sizeof(C) == (BLOCK + 2 * GROUP) * 4, "the stores below cover the slots");
#pragma unroll
	for (int k = 0; k < VPQ / 4; k++)
		*reinterpret_cast<uint4 *>(&stage[sub * VPQ + 4 * k]) = make_uint4(0, 0, 0, 0);
	*reinterpret_cast<uint2 *>(&stage[BLOCK + 2 * sub]) = make_uint2(0, 0);
	// bytes of a varint: ceil(L / 7) = (37 L + 219) >> 8 for a bit length L in 1..32
	auto nbytes = [](uint32_t x) { return (uint32_t(__umul24(bitlen(x | 1u), 37u)) + 219u) >> 8; };
	uint32_t count = 0;
#pragma unroll
	for (int i = 0; i < VPQ; i++) {
		u[i] = enc == OROCH_VARINT ? zz<E>(u[i]) : C((u[i] - origin) & E::MASK);
		count += nbytes(u[i]);
	}
	count = vok ? count : 0u;
	const uint32_t at = row_scan_sum(count) - count; // my first byte
	wave_sync();
	if (vok) {
		uint32_t word = at >> 2, fill = at & 3u; // dword under way, bytes it has
		uint64_t acc = 0;
#pragma unroll
		for (int i = 0; i < VPQ; i++) {
			const uint32_t x = u[i];
			const uint32_t nby = nbytes(x);
			// the seven-bit groups a byte apart, continuation bits on all but the last
			const uint32_t lo = (x & 0x7fu) | ((x << 1) & 0x7f00u) | ((x << 2) & 0x7f0000u) |
					    ((x << 3) & 0x7f000000u);
			const uint32_t cont = 0x80808080u & ((nby > 4u ? 0u : 1u << (8u * (nby - 1u))) - 1u);
			uint64_t w = lo | cont;
			if constexpr (E::W > 28)
				w |= uint64_t(x >> 28) << 32;
			acc |= w << (8u * fill);
			fill += nby;
			if (fill >= 4u) {
				atomicOr(&stage[word++], uint32_t(acc));
				acc >>= 32;
				fill -= 4u;
			}
			if constexpr (E::W > 21) {
				// (three bytes under way and a five-byte value make eight)
				if (fill >= 4u) {
					atomicOr(&stage[word++], uint32_t(acc));
					acc >>= 32;
					fill -= 4u;
				}
			}
		}
		if (fill)
			atomicOr(&stage[word], uint32_t(acc));
	}
	wave_sync();
	if (vok) {
		// metadata, integer_codec.h:128-151: [enc] or [enc][origin varint], zeros up to byte 8
		if (sub == 0) {
			const uint32_t zo = zz<E>(origin);
			const uint32_t nby = nbytes(zo);
			const uint32_t lo = (zo & 0x7fu) | ((zo << 1) & 0x7f00u) | ((zo << 2) & 0x7f0000u) |
					    ((zo << 3) & 0x7f000000u);
			const uint32_t cont = 0x80808080u & ((nby > 4u ? 0u : 1u << (8u * (nby - 1u))) - 1u);
			uint64_t hd = (uint64_t(zo >> 28) << 32) | lo | cont;
			hd = enc == OROCH_VARFOR ? hd << 8 : 0;
			*reinterpret_cast<uint2 *>(out) = make_uint2(uint32_t(hd) | enc, uint32_t(hd >> 32));
		}
		const uint8_t *bytes = reinterpret_cast<const uint8_t *>(stage);
		for (uint32_t k = uint32_t(sub) * 8u; k < size - doff; k += GROUP * 8u)
			*reinterpret_cast<uint2 *>(out + doff + k) = *reinterpret_cast<const uint2 *>(bytes + k);
	}
	wave_sync();
	return todo;
}

// The kernel is latency-bound (dependent LDS round trips per quad), so it is
// shaped for occupancy: registers capped for six wavefronts per SIMD, and the
// one-block path's image shares LDS with the slots (the two are never live
// together).
template <typename E>
__global__ __launch_bounds__(WAVE *WAVES_PER_WG) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_encode_quad(const typename E::S *__restrict__ src,
								      size_t n, size_t nblocks,
								      const oroch_block_desc *__restrict__ desc,
								      const uint64_t *__restrict__ offsets,
								      uint8_t *__restrict__ dst)
{
	using C = typename E::C;
	__shared__ __attribute__((aligned(16))) C slotss[WAVES_PER_WG][QUAD * SLOT_STRIDE];
	__shared__ __attribute__((aligned(16))) uint8_t tailss[WAVES_PER_WG][QUAD * TAIL_BYTES];
	__shared__ __attribute__((aligned(16))) uint8_t headss[WAVES_PER_WG][QUAD * 16];
	static_assert(sizeof(C) * QUAD * SLOT_STRIDE >= size_t(image_bytes<E>()), "image must fit the slots");
	__shared__ uint4 wt[65];
	const int wave = uniform(int(threadIdx.x / WAVE)), lane = lane_id();
	load_width_table(wt);
	uint32_t *image = reinterpret_cast<uint32_t *>(slotss[wave]);
	const size_t nquads = (nblocks + QUAD - 1) / QUAD;
	const size_t stride = size_t(gridDim.x) * WAVES_PER_WG;
	for (size_t q = size_t(blockIdx.x) * WAVES_PER_WG + wave; q < nquads; q += stride) {
		const size_t blk0 = q * QUAD;
		uint32_t done = 0;
		if (blk0 + QUAD <= nblocks && (blk0 + QUAD) * BLOCK <= n) {
			// (an opaque lane number: the address arithmetic stays in the loop
			// instead of in registers that live across it)
			int ln = lane;
			asm volatile("" : "+v"(ln));
			const size_t blk = blk0 + size_t(uint32_t(ln) >> 4);
			C v[VPQ];
			load16<E>(src + blk * BLOCK + size_t(uint32_t(ln) & 15u) * VPQ, v);
			const uint4 *dq = reinterpret_cast<const uint4 *>(desc + blk);
			const uint4 da = dq[0], db = dq[1];
			const uint64_t off = offsets[blk];
			done = encode_quad_fast<E>(v, da, db, off, lane, slotss[wave], tailss[wave], headss[wave],
						   wt, dst);
		}
		if (done != 15u) {
			if (blk0 + QUAD <= nblocks && (blk0 + QUAD) * BLOCK <= n)
				done |= encode_quad_varints<E>(src, desc, offsets, blk0, lane, slotss[wave], dst);
			for (int b = 0; b < QUAD && blk0 + size_t(b) < nblocks; b++) {
				if ((done >> b) & 1u)
					continue;
				const size_t blk = blk0 + size_t(b);
				const int nv = block_count(n, blk);
				C v[VPL];
				load_block<E>(src, blk * BLOCK, nv, lane, v);
				selection<C> sel;
				get_desc(&desc[blk], sel);
				const uint64_t off = uniform(offsets[blk]);
				const uint32_t len = nv == BLOCK
							     ? encode_block<E, true>(v, nv, lane, sel, wt, image)
							     : encode_block<E, false>(v, nv, lane, sel, wt, image);
				flush_image(image, len, dst + off, lane);
				wave_sync();
			}
		}
		wave_sync();
	}
}

} // namespace oroch_hip
