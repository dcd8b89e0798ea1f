// octo.hpp -- selection, offsets and encoding in ONE launch, eight consecutive
// full blocks per wavefront, eight lanes per block: the values are read from
// HBM once (integer_group.h:44-60 for every block of the array;
// integer_codec.h:234-402, integer_stats.h:48-53, 74-80).
//
// Lane (g, sub) holds values 32 sub .. 32 sub + 31 of block g of the octet.
// What is uniform per block (read-out of the histogram, the basic encodings,
// the candidate widths, header and lists) lives in the vector registers of the
// block's eight lanes, so one instruction stream serves eight blocks; each lane
// looks after four candidate widths, 4 sub + 1 .. 4 sub + 4.  Reductions inside
// a block are three DPP steps (two quad permutes and a mirror of the half
// row); a DPP row holds two blocks, so scans along a row carry the two blocks
// in separate bit fields (sums) or on separate number ranges (maxima).
//
// Included by group.hip after select_quad.hpp and encode_quad.hpp; 8-, 16- and
// 32-bit element types.
#pragma once

namespace oroch_hip {

constexpr int OCT = 8;          // blocks per wavefront step
constexpr int OG = WAVE / OCT;  // lanes per block
constexpr int VPO = BLOCK / OG; // values per lane and block
static_assert(OG == 8 && VPO == 32, "lane arithmetic below");

// Histogram: bin c + 1 counts deltas with c leading zeros (bin 0: zero deltas),
// OH_COPIES words per bin, one per lane number modulo OH_COPIES.
#ifndef OH_COPIES
#define OH_COPIES 4
#endif
static_assert(OH_COPIES == 4, "bins are read as one 16-byte word");
constexpr int OH_DWORDS = 140;  // 33 bins x 4 copies, padded: the tables of the eight blocks start 12 banks apart
constexpr int ON_DWORDS = 64;   // per block: exception counts by width (entry b - 1), zero above the widest
constexpr int OP_DWORDS = 32;   // per block: candidate parameters by width (entry b - 1)
constexpr int OT_BYTES = 256;   // exception lists + padding a block stages in LDS
constexpr int OCT_SLOT_STRIDE = BLOCK + FIELDS_PAD;

template <typename V, typename Op>
__device__ __forceinline__ V oct_allreduce(V x, V ident, Op op)
{
	x = op(x, dpp_mov<0xb1>(ident, x));
	x = op(x, dpp_mov<0x4e>(ident, x));
	x = op(x, dpp_mov<0x141>(ident, x)); // lane i of a half row <-> lane 7 - i
	return x;
}
__device__ __forceinline__ uint32_t oct_sum(uint32_t x)
{
	return oct_allreduce(x, 0u, [](uint32_t a, uint32_t b) { return a + b; });
}
__device__ __forceinline__ uint32_t oct_max(uint32_t x)
{
	return oct_allreduce(x, 0u, [](uint32_t a, uint32_t b) { return a > b ? a : b; });
}
__device__ __forceinline__ uint32_t oct_min(uint32_t x)
{
	return oct_allreduce(x, ~0u, [](uint32_t a, uint32_t b) { return a < b ? a : b; });
}
__device__ __forceinline__ uint32_t oct_or(uint32_t x)
{
	return oct_allreduce(x, 0u, [](uint32_t a, uint32_t b) { return a | b; });
}

// What depends only on the lane number.
struct oct_lane
{
	uint32_t g, sub;
	uint32_t sh16;   // 16 in the upper block of a DPP row, 0 in the lower
	uint32_t hi256;  // 256 in the upper block of a DPP row, 0 in the lower
};
__device__ __forceinline__ oct_lane make_oct_lane(int lane)
{
	oct_lane l;
	l.g = uint32_t(lane) >> 3;
	l.sub = uint32_t(lane) & 7u;
	l.sh16 = (uint32_t(lane) & 8u) << 1;
	l.hi256 = (uint32_t(lane) & 8u) << 5;
	return l;
}

// Inclusive prefix sum over the lanes of a block; x and every sum below 2^16.
__device__ __forceinline__ uint32_t oct_scan_sum(uint32_t x, const oct_lane &l)
{
	x <<= l.sh16;
	x += dpp_mov<0x111>(0u, x);
	x += dpp_mov<0x112>(0u, x);
	x += dpp_mov<0x114>(0u, x);
	x += dpp_mov<0x118>(0u, x);
	return (x >> l.sh16) & 0xffffu;
}

// p1: 0, or a position of this lane's block counted from 1.  Returns the
// largest p1 among the lower lanes of the block (0 if none).
__device__ __forceinline__ uint32_t oct_prev(uint32_t p1, const oct_lane &l)
{
	// the upper block of a row works 256 higher, so that whatever reaches it
	// from the lower block falls away in the saturating subtraction
	uint32_t y = p1 ? p1 + l.hi256 : 0u;
	y = row_scan_max(y);
	y = dpp_mov<0x111>(0u, y);
	asm volatile("" : "+v"(y)); // the shift stays an instruction of its own (see select_quad)
	return __builtin_elementwise_sub_sat(y, l.hi256);
}

// ---- selection --------------------------------------------------------------------

// hq: OCT * OH_DWORDS, no: OCT * ON_DWORDS, pw: OCT * OP_DWORDS dwords of LDS
// owned by this wavefront.  v[] holds the lane's 32 raw values on entry and
// their deltas from the block's minimum on return; mn is that minimum, msel
// the lane's exceptions (bit i for value i) when a patched encoding was chosen.
template <typename E>
__device__ __forceinline__ void select_octo(typename E::C (&v)[VPO], int lane, uint32_t *hq, uint32_t *no,
					    uint32_t *pw, const uint4 *wt, quad_selection &sel,
					    typename E::C &mn_out, uint32_t &msel)
{
	using C = typename E::C;
	static_assert(sizeof(C) == 4, "element types up to 32 bits");
	asm volatile("" : "+v"(lane)); // lane arithmetic is recomputed here, not carried through the callers' loops
	const oct_lane l = make_oct_lane(lane);
	const uint32_t g = l.g, sub = l.sub;
	const uint32_t b0 = 4u * sub + 1u; // this lane's widths: b0 .. b0 + 3

	sel.encoding = OROCH_NORMAL;
	sel.nbits = 0;
	sel.metaspace = 1;
	sel.dataspace = 0;
	sel.noutliers = 0;
	sel.ienc = sel.venc = OROCH_NORMAL;
	sel.inbits = sel.vnbits = 0;
	sel.origin = 0;
	msel = 0;

	// integer_stats.h:88-95: minimum and maximum (signed compare for int32,
	// else on order-preserving unsigned keys)
	C mn, mx;
	if constexpr (E::sgn && E::W == 32) {
		int32_t smin = int32_t(v[0]), smax = int32_t(v[0]);
#pragma unroll
		for (int i = 1; i < VPO; i++) {
			smin = int32_t(v[i]) < smin ? int32_t(v[i]) : smin;
			smax = int32_t(v[i]) > smax ? int32_t(v[i]) : smax;
		}
		mn = C(oct_min(uint32_t(smin) ^ 0x80000000u) ^ 0x80000000u);
		mx = C(oct_max(uint32_t(smax) ^ 0x80000000u) ^ 0x80000000u);
	} else {
		C kmin = v[0] ^ E::FLIP, kmax = kmin;
#pragma unroll
		for (int i = 1; i < VPO; i++) {
			const C k = v[i] ^ E::FLIP;
			kmin = k < kmin ? k : kmin;
			kmax = k > kmax ? k : kmax;
		}
		mn = oct_min(kmin) ^ E::FLIP;
		mx = oct_max(kmax) ^ E::FLIP;
	}
	mn_out = mn;
	const C zmn = zz<E>(mn), zmx = zz<E>(mx);
	const uint32_t vo = vsp(zmn);
	const bool constant = mn == mx; // integer_codec.h:255-264, applied at the end
	// widest delta and widest zigzag image (integer_codec.h:441-476)
	const uint32_t nbmax = bitlen(uint32_t((mx - mn) & E::MASK));
	const uint32_t nbz0 = bitlen(uint32_t(zmn > zmx ? zmn : zmx));
	const uint32_t nbz = nbz0 ? nbz0 : 1u;

	// Bit-length histogram of v - min, integer_stats.h:97-102.
	constexpr int NC = OH_COPIES;
	uint32_t *table = hq + mul24(g, OH_DWORDS);
#pragma unroll
	for (int k = 0; k < (OH_DWORDS + OG * 4 - 1) / (OG * 4); k++)
		if ((k * OG + int(sub)) * 4 < OH_DWORDS)
			*reinterpret_cast<uint4 *>(&table[(k * OG + sub) * 4]) = make_uint4(0, 0, 0, 0);
	uint32_t *nog = no + g * ON_DWORDS;
	*reinterpret_cast<uint4 *>(&nog[32 + 4 * sub]) = make_uint4(0, 0, 0, 0);
	wave_sync();
	uint32_t *mine = table + NC + (sub & uint32_t(NC - 1));
	// The zigzag images need no histogram: what selection wants of them is the
	// sum of their varint lengths.  A value whose zigzag image is L bits long
	// takes ceil(L / 7) = (37 L + 219) >> 8 bytes (L in 1..32; zero takes one
	// byte like L = 1), and L is ZL minus the count of redundant leading bits,
	// clamped for the "none at all" answer.
	constexpr uint32_t ZL = E::sgn ? 33u : 32u, ZCAP = ZL - 1u;
	uint32_t zbytes = 0;
#pragma unroll
	for (int i = 0; i < VPO; i++) {
		int cz;
		if constexpr (E::sgn)
			cz = leading_sign_bits(int32_t(v[i] << (32 - E::W)) >> (32 - E::W));
		else
			cz = leading_zeros(uint32_t(v[i]));
		const C d = C((v[i] - mn) & E::MASK);
		const int cd = leading_zeros(uint32_t(d));
		atomicAdd(&mine[cd * NC], 1u);
		const uint32_t czc = uint32_t(cz) < ZCAP ? uint32_t(cz) : ZCAP; // -1 reads as huge
		zbytes += (37u * ZL + 219u - mul24(czc, 37u)) >> 8;
		v[i] = d;
	}
	wave_sync();
	// width b <-> bin 33 - b: this lane's four bins lie next to each other
	uint32_t h[4];
	{
		const uint32_t *at = &table[(29u - 4u * sub) * NC];
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const uint4 q = *reinterpret_cast<const uint4 *>(at + (3 - r) * NC);
			h[r] = q.x + q.y + q.z + q.w;
		}
	}
	uint32_t zero_d;
	{
		const uint4 q = *reinterpret_cast<const uint4 *>(table);
		zero_d = q.x + q.y + q.z + q.w;
	}

	// values longer than b bits, for this lane's four widths
	uint32_t nout[4];
	{
		const uint32_t mine4 = h[0] + h[1] + h[2] + h[3];
		const uint32_t upto = oct_scan_sum(mine4, l);
		nout[3] = uint32_t(BLOCK) - zero_d - upto;
		nout[2] = nout[3] + h[3];
		nout[1] = nout[2] + h[2];
		nout[0] = nout[1] + h[1];
	}
	*reinterpret_cast<uint4 *>(&nog[4 * sub]) = make_uint4(nout[0], nout[1], nout[2], nout[3]);

	// the two varint footprints, integer_codec.h:483-489: the deltas' from the
	// histogram, the zigzag images' from the sum above
	uint32_t vfs, vis;
	{
		uint32_t part = zbytes << 16;
#pragma unroll
		for (int r = 0; r < 4; r++)
			part += mul24(h[r], mul24(b0 + uint32_t(r) + 6u, 37u) >> 8);
		const uint32_t t = oct_sum(part);
		vfs = (t & 0xffffu) + zero_d;
		vis = t >> 16;
	}

	// select_basic, integer_codec.h:422-496; strictly smaller replaces (:413)
	uint32_t best = BLOCK * E::BYTES, bmeta0 = 0;
	{
		const uint32_t total = wt[nbz].w + 1u;
		if (total < best) {
			best = total;
			bmeta0 = 1;
			sel.encoding = OROCH_BITPCK;
			sel.nbits = nbz;
		}
	}
	{
		const uint32_t total = wt[nbmax ? nbmax : 1u].w + 1u + vo;
		if (total < best) {
			best = total;
			bmeta0 = 1 + vo;
			sel.encoding = OROCH_BITFOR;
			sel.nbits = nbmax;
			sel.origin = mn;
		}
		if (vis < best) {
			best = vis;
			bmeta0 = 0;
			sel.encoding = OROCH_VARINT;
			sel.nbits = 0;
			sel.origin = 0;
		}
		if (vfs + vo < best) {
			best = vfs + vo;
			bmeta0 = vo;
			sel.encoding = OROCH_VARFOR;
			sel.nbits = 0;
			sel.origin = mn;
		}
	}
	sel.metaspace = 1 + bmeta0;
	sel.dataspace = best - bmeta0;

	// ---- patched frame of reference, integer_codec.h:280-383 ----
	wave_sync();
	const uint32_t basic_meta = 1 + vo;
	uint32_t *pwg = pw + g * OP_DWORDS;
	// est[r]: the estimate of width b0 + r (:327-335), ~0 where the width is no
	// candidate; low[r]: what the width needs at the very least
	uint32_t est[4], low[4];
	{
		uint32_t par[4];
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const uint32_t b = b0 + uint32_t(r);
			const bool active = b < nbmax && h[r] != 0 && !constant;
			// varint bytes of the high parts: each value of bit length k > b
			// costs ceil((k - b) / 7), i.e. the sum over j >= 0 of nout(b + 7j)
			// (:308-312)
			uint32_t valvar = nout[r] + nog[b + 6] + nog[b + 13];
			if constexpr (E::W > 16)
				valvar += nog[b + 20] + nog[b + 27];
			const uint32_t xmeta = 2 + (nout[r] < 128u ? 1u : 2u);
			const uint4 cm = wt[active ? nbmax - b : 1u];
			const uint32_t valpck = pck_bytes(nout[r], cm);
			const bool vpck = valpck < valvar;
			const uint32_t vdata = vpck ? valpck : valvar;
			const uint32_t indmin = nout[r] > 128u ? 32u : 16u; // min(pck(nout, 1), 256), nout in 1..255
			const uint32_t fixed = basic_meta + xmeta + wt[b].w + vdata;
			// (a width whose fixed part alone fills a raw block never passes the gate)
			const bool cand = active && fixed < 1024u;
			est[r] = cand ? fixed + indmin : ~0u;
			low[r] = fixed + (nout[r] < indmin ? nout[r] : indmin); // a varint list of one-byte gaps
			par[r] = (fixed & 0x7ffu) | (nout[r] << 11) | (vpck ? 0x80000000u : 0u);
		}
		*reinterpret_cast<uint4 *>(&pwg[4 * sub]) = make_uint4(par[0], par[1], par[2], par[3]);
	}
	wave_sync();

	// One candidate width of each block, all eight blocks in step: the index
	// list of width bL (integer_codec.h:338-351) and what the block then takes.
	auto evaluate = [&](uint32_t bL, uint32_t par, uint32_t &required, uint32_t &how, uint32_t &m) {
		const uint32_t fixedL = par & 0x7ffu, noutL = (par >> 11) & 0x1ffu;
		// my exceptions at this width, bit i for value i: d >= 2^bL exactly when
		// d + (2^32 - 2^bL) carries
		const uint32_t lift = 0u - (1u << (bL & 31u));
		m = 0;
#pragma unroll
		for (int i = VPO - 1; i >= 0; i--) {
			uint32_t sum;
			const uint32_t carry = __builtin_add_overflow(uint32_t(v[i]), lift, &sum) ? 1u : 0u;
			m = m + m + carry;
		}
		// Index gaps, :338-351.  The gap in front of my first exception reaches
		// back to the last exception of the lanes before me; the gaps between
		// my own exceptions are runs of zeros inside m.
		const uint32_t hb = 31u - uint32_t(__clz(m | 1u));
		const uint32_t fb = uint32_t(__builtin_ctz(m | 0x80000000u));
		const uint32_t prev1 = oct_prev(m ? sub * VPO + hb + 1u : 0u, l);
		const uint32_t gfirst = m ? sub * VPO + fb - prev1 : 0u;
		// zeros strictly between my lowest and highest exception
		const uint32_t z1 = ~m & ((2u << hb) - 1u) & ~((1u << fb) - 1u);
		const uint32_t z2 = z1 & (z1 >> 1), z4 = z2 & (z2 >> 2), z8 = z4 & (z4 >> 4), z16 = z8 & (z8 >> 8);
		const uint32_t inbits = (z1 < 1u ? z1 : 1u) + (z2 < 1u ? z2 : 1u) + (z4 < 1u ? z4 : 1u) +
					(z8 < 1u ? z8 : 1u) + (z16 < 1u ? z16 : 1u);
		const uint32_t fbits = bitlen(gfirst);
		uint32_t indnbits = oct_max(fbits > inbits ? fbits : inbits); // bit length of the longest gap
		// 256 positions hold at most one gap of 128 or more
		const uint32_t indvar = noutL + (indnbits == 8u ? 1u : 0u);
		indnbits = indnbits ? indnbits : 1u;
		const uint4 ci = wt[indnbits];
		const uint32_t indpck = pck_bytes(noutL, ci);
		const bool ipck = indpck < indvar;
		required = fixedL + (ipck ? indpck : indvar);
		how = bL | (ipck ? indnbits << 8 : 0u) | ((ipck ? uint32_t(OROCH_BITPCK) : uint32_t(OROCH_VARINT)) << 16);
	};

	uint32_t got = 0, gotpar = 0;
	// The reference walks the widths upwards; a width is looked at when its
	// estimate is below the best total so far (:327-335) and adopted when what
	// it really needs is below it too (:353-383).  Widths whose estimate is
	// not below the best of the basic encodings never get looked at.  Of the
	// others, take the one whose lower bound (the index list at its
	// shortest) is lowest and evaluate it: if both its estimate and its real
	// need are below every other width's lower bound (and below the basic
	// best), the walk adopts exactly this width -- nothing before it can have
	// pushed the best total low enough to hide it, nothing after it can beat
	// it -- and one evaluation has replaced the walk.  Otherwise the walk runs.
	bool walk;
	{
		uint32_t key = ~0u;
#pragma unroll
		for (int r = 0; r < 4; r++) {
			est[r] = est[r] < best ? est[r] : ~0u;
			const uint32_t k = est[r] != ~0u ? (low[r] << 6) | (b0 + uint32_t(r) - 1u) : ~0u;
			low[r] = k; // from here on the key
			key = k < key ? k : key;
		}
		const uint32_t first = oct_min(key);
		uint32_t rest = ~0u;
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const uint32_t k = low[r] == first ? ~0u : low[r];
			rest = k < rest ? k : rest;
		}
		const uint32_t second = oct_min(rest);
		const bool any = first != ~0u;
		const uint32_t bL = any ? (first & 63u) + 1u : 1u;
		const uint32_t par = pwg[bL - 1u];
		uint32_t required, how, m;
		evaluate(bL, par, required, how, m);
		const uint32_t indmin = ((par >> 11) & 0x1ffu) > 128u ? 32u : 16u;
		const uint32_t e = (par & 0x7ffu) + indmin;
		const uint32_t top = required > e ? required : e;
		const uint32_t bound = best < (second >> 6) ? best : (second >> 6); // second >> 6 is huge when there is none
		const bool settled = top < bound;
		if (any && settled) {
			best = required;
			got = how;
			gotpar = par;
			msel = m;
		}
		walk = any && !settled;
	}
	if (__ballot(walk) != 0) {
		// Replay of the reference's ascending loop for the blocks that need it
		// (the others idle through): each block takes its narrowest remaining
		// candidate.  `best` only shrinks, so a width whose estimate is not
		// below it fails the gate now and ever after: each round a block moves
		// straight on to its narrowest remaining width that passes.
#pragma unroll
		for (int r = 0; r < 4; r++)
			est[r] = walk ? est[r] : ~0u;
		for (;;) {
			// bit b - 1 for every width of the block still in the running
			uint32_t nib = 0;
#pragma unroll
			for (int r = 3; r >= 0; r--)
				nib = nib + nib + (est[r] < best ? 1u : 0u);
			const uint32_t cands = oct_or(nib << (4u * sub));
			if (__ballot(cands != 0) == 0)
				break;
			const bool tryit = cands != 0;
			const uint32_t bL = tryit ? uint32_t(__builtin_ctz(cands)) + 1u : 1u;
#pragma unroll
			for (int r = 0; r < 4; r++)
				est[r] = b0 + uint32_t(r) == bL ? ~0u : est[r];
			const uint32_t par = pwg[bL - 1u];
			uint32_t required, how, m;
			evaluate(bL, par, required, how, m);
			if (tryit && required < best) {
				best = required;
				got = how;
				gotpar = par;
				msel = m;
			}
		}
	}
	if (got) {
		// the last width adopted, :365-382
		const uint32_t bL = got & 0xffu, nx = (gotpar >> 11) & 0x1ffu;
		const uint32_t xm = 2 + (nx < 128u ? 1u : 2u);
		sel.encoding = OROCH_BITPFR;
		sel.origin = mn;
		sel.nbits = bL;
		sel.noutliers = nx;
		sel.venc = (gotpar >> 31) ? OROCH_BITPCK : OROCH_VARINT;
		sel.vnbits = nbmax - bL;
		sel.ienc = got >> 16;
		sel.inbits = (got >> 8) & 0xffu;
		sel.metaspace = 1 + basic_meta + xm;
		sel.dataspace = best - basic_meta - xm;
	}
	if (constant) {
		sel.encoding = OROCH_NAUGHT;
		sel.nbits = 0;
		sel.origin = mn;
		sel.metaspace = 1 + vo;
		sel.dataspace = 0;
		sel.noutliers = 0;
		sel.ienc = sel.venc = OROCH_NORMAL;
		sel.inbits = sel.vnbits = 0;
		msel = 0;
	}
}

// ---- encoding ---------------------------------------------------------------------

// The blocks of an octet that share a width form a class (an 8-bit mask,
// uniform); `members` lists them in ascending order, four bits each.
struct oct_class
{
	uint32_t count, members;
};
__device__ __forceinline__ oct_class make_oct_class(uint32_t mask)
{
	oct_class c;
	c.count = uint32_t(__builtin_popcount(mask));
	c.members = 0;
	uint32_t at = 0;
	for (uint32_t m = mask; m; m &= m - 1u, at += 4)
		c.members |= uint32_t(__builtin_ctz(m)) << at;
	return c;
}

// One bit per block from a ballot of a block-uniform condition.
__device__ __forceinline__ uint32_t oct_mask(uint64_t ballot)
{
	// bit 8 g of the ballot to bit g
	const uint64_t x = ballot & 0x0101010101010101ull;
	return uint32_t((x * 0x0102040810204080ull) >> 56);
}

// Lane p, p + 64, ... packs word p of a class of blocks (word j of the class's
// rank-th block) from the slots and stores it at its place in the stream.
template <typename E, int NB>
__device__ __forceinline__ void pack_oct_words(const oct_class &c, const typename E::C *slots, uint32_t data_at,
						int lane, uint8_t *out)
{
	using C = typename E::C;
	constexpr uint32_t NBLK = (BLOCK + 128 / NB - 1) / (128 / NB);
	constexpr uint32_t RECIP = (65536u + NBLK - 1u) / NBLK; // floor(p / NBLK) == (p * RECIP) >> 16 for p < 8 * NBLK <= 512
	const uint32_t P = c.count * NBLK;
	asm volatile("" : "+v"(lane)); // keep lane arithmetic of every width out of the callers' loops
#pragma unroll
	for (uint32_t r = 0; r < (OCT * NBLK + WAVE - 1) / WAVE; r++) {
		const uint32_t p = uint32_t(lane) + r * WAVE;
		if (r * WAVE >= P)
			break;
		const uint32_t rank = mul24(p, RECIP) >> 16;
		const uint32_t j = p - mul24(rank, NBLK);
		const uint32_t gp = (c.members >> ((rank & 7u) * 4u)) & 7u;
		const uint32_t at = lane_gather(data_at, int(gp * OG));
		if (p < P) {
			uint32_t w[4];
			pack_fields<C, NB, true>(slots + mul24(gp, OCT_SLOT_STRIDE), int(j), w);
			*reinterpret_cast<uint4 *>(out + at + j * 16u) = make_uint4(w[0], w[1], w[2], w[3]);
		}
	}
}

template <typename E, int LO, int HI>
__device__ __forceinline__ void pack_oct_dispatch(uint32_t nb, const oct_class &c, const typename E::C *slots,
						  uint32_t data_at, int lane, uint8_t *out)
{
	if constexpr (LO == HI) {
		pack_oct_words<E, LO>(c, slots, data_at, lane, out);
	} else {
		constexpr int MID = (LO + HI) / 2;
		if (nb <= uint32_t(MID))
			pack_oct_dispatch<E, LO, MID>(nb, c, slots, data_at, lane, out);
		else
			pack_oct_dispatch<E, MID + 1, HI>(nb, c, slots, data_at, lane, out);
	}
}

// What the two halves of the encoder hand each other (block-uniform, one copy per lane).
struct oct_plan
{
	bool ok;            // this block goes through the eight-block path
	uint32_t fast;      // the blocks of the octet that do, one bit each (uniform)
	uint32_t nb, doff, list_at, size;
};

// First half, for which the octet's place in the stream need not be known: the
// deltas (or zigzag images) go to the slots, header and exception lists are
// laid out in LDS (heads: 16 bytes per block, tails: OT_BYTES per block).
// d[]: deltas from mn as select_octo leaves them; m: the lane's exceptions.
template <typename E>
__device__ __forceinline__ oct_plan encode_octo_prepare(const typename E::C (&d)[VPO], typename E::C mn,
							 const quad_selection &sel, uint32_t m, int lane,
							 typename E::C *slots, uint8_t *tails, uint8_t *heads,
							 const uint4 *wt)
{
	using C = typename E::C;
	static_assert(sizeof(C) == 4, "element types up to 32 bits");
	asm volatile("" : "+v"(lane));
	const oct_lane l = make_oct_lane(lane);
	const uint32_t g = l.g, sub = l.sub;

	const uint32_t enc = sel.encoding, nb = sel.nbits, nout = sel.noutliers;
	const uint32_t ienc = sel.ienc, inb = sel.inbits, venc = sel.venc, vnb = sel.vnbits;
	oct_plan plan;
	plan.nb = nb;
	plan.size = stream_size(sel);
	plan.doff = (sel.metaspace + 7u) & ~7u; // integer_group.h:50-52
	const bool width_ok = nb >= uint32_t(FIELDS_MIN_NB) && nb <= uint32_t(E::W);
	const uint32_t pck = wt[width_ok ? nb : 1u].w;
	plan.list_at = plan.doff + pck;
	plan.ok = (enc == OROCH_BITPCK || enc == OROCH_BITFOR || enc == OROCH_BITPFR) && width_ok &&
		  sel.dataspace >= pck && plan.size >= plan.list_at && plan.size - plan.list_at <= uint32_t(OT_BYTES);
	plan.fast = oct_mask(__ballot(plan.ok));
	if (plan.fast == 0)
		return plan;
	const bool ok = plan.ok;
	const bool has_exc = ok && enc == OROCH_BITPFR;

	// deltas (or zigzag images) of all values into the slots, in full: the
	// packer cuts them to the width
	C *mine = slots + mul24(g, OCT_SLOT_STRIDE);
	const bool zig = E::sgn && enc == OROCH_BITPCK;
	if (__ballot(enc == OROCH_BITPCK) != 0) {
		// bitpck packs the values themselves (zigzag images of signed ones)
#pragma unroll
		for (int k = 0; k < VPO / 4; k++) {
			C u[4];
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const C raw = C((d[4 * k + i] + mn) & E::MASK);
				u[i] = enc == OROCH_BITPCK ? (zig ? zz<E>(raw) : raw) : d[4 * k + i];
			}
			*reinterpret_cast<uint4 *>(&mine[sub * VPO + 4 * k]) = make_uint4(u[0], u[1], u[2], u[3]);
		}
	} else {
#pragma unroll
		for (int k = 0; k < VPO / 4; k++)
			*reinterpret_cast<uint4 *>(&mine[sub * VPO + 4 * k]) =
				make_uint4(d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3]);
	}
	// the last packed word of a block reads past value 255: zeros there
	*reinterpret_cast<uint4 *>(&mine[BLOCK + 4 * sub]) = make_uint4(0, 0, 0, 0);
	// exception list staging starts out as zeros (it also supplies the padding)
	uint8_t *tail = tails + g * OT_BYTES;
	*reinterpret_cast<uint4 *>(tail + 32 * sub) = make_uint4(0, 0, 0, 0);
	*reinterpret_cast<uint4 *>(tail + 32 * sub + 16) = make_uint4(0, 0, 0, 0);
	m = has_exc ? m : 0u;

	// metadata, integer_codec.h:128-151, 177-182, 196-204: lane `sub` makes
	// byte `sub` of the (at most 11-byte) header, and byte `sub + 8` where a
	// header is that long
	{
		const bool has_o = enc != OROCH_BITPCK;
		const C zo = zz<E>(sel.origin);
		uint32_t olen = (bitlen(zo) + 6u) / 7u;
		olen = has_o ? (olen ? olen : 1u) : 0u;
		const uint32_t p_nb = 1u + olen, p_nout = p_nb + 1u;
		const uint32_t nlen = nout >= 128u ? 2u : 1u;
		const uint32_t p_i = p_nout + nlen, p_v = p_i + 1u;
		auto header_byte = [&](uint32_t pos) -> uint32_t {
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
			return byte;
		};
		uint8_t *head = heads + g * 16;
		head[sub] = uint8_t(header_byte(sub));
		if (__ballot(ok && plan.doff > 8u) != 0)
			head[sub + 8] = uint8_t(header_byte(sub + 8u));
	}
	wave_sync();

	// exception lists, integer_codec.h:563-581: gap-coded positions
	// (offset.h:55-60) then high parts, bit-packed or as varints
	if (__ballot(has_exc) != 0) {
		const uint32_t cnt = uint32_t(__popc(m));
		const uint32_t before = oct_scan_sum(cnt, l) - cnt;
		const uint32_t hb = 31u - uint32_t(__clz(m | 1u));
		const uint32_t fb = uint32_t(__builtin_ctz(m | 0x80000000u));
		// position + 1 of the last exception in lower lanes (0: none)
		const uint32_t prev1 = oct_prev(m ? sub * VPO + hb + 1u : 0u, l);
		const uint4 iw = wt[inb ? inb : 1u], vw = wt[vnb ? vnb : 1u];
		// first pass over my exceptions: how many list bytes they take.  A
		// gap takes two varint bytes from 128 on, which only the gap in front
		// of a lane's first exception can reach.
		const uint32_t gfirst = sub * VPO + fb - prev1;
		// (a block on this path has at most OT_BYTES of lists, so with both
		// lists present either takes fewer than 256 bytes; a bit-packed index
		// list is sized from the count alone)
		const uint32_t ib = ienc == OROCH_BITPCK ? 0u : cnt + (m && gfirst >= 128u ? 1u : 0u);
		uint32_t vb = 0;
		if (__ballot(has_exc && venc != OROCH_BITPCK) != 0) {
			// (only where the high parts are varints: the byte counts of two
			// blocks share a register in the scan below, and those of a block
			// with a long bit-packed list would spill over)
			uint32_t mm = venc != OROCH_BITPCK ? m : 0u;
			while (__ballot(mm != 0) != 0) {
				if (mm) {
					const uint32_t pos = sub * VPO + uint32_t(__builtin_ctz(mm));
					vb += vsp(C(mine[pos] >> nb));
					mm &= mm - 1u;
				}
			}
		}
		// (both sums through one scan: the index bytes in the low half)
		const uint32_t both = oct_scan_sum(ib | (vb << 8), l);
		const uint32_t isum = both & 0xffu, vsum = both >> 8;
		const uint32_t itotal = oct_max(isum);
		// (bitpck.h:61-71 with the 16-bit reciprocal: nout + capacity - 1 < 512)
		const uint32_t ilen = ienc == OROCH_BITPCK ? (mul24(nout + iw.x - 1u, iw.z) >> 16) << 4 : itotal;
		uint32_t io = isum - ib, vo = ilen + vsum - vb, r = before;
		uint32_t *tail32 = reinterpret_cast<uint32_t *>(tail);
		{
			uint32_t mm = m;
			int prev = int(prev1) - 1;
			while (__ballot(mm != 0) != 0) {
				if (mm) {
					const int pos = int(sub * VPO) + int(__builtin_ctz(mm));
					const uint32_t gap = uint32_t(pos - prev - 1);
					prev = pos;
					const C hi = mine[pos] >> nb;
					if (ienc == OROCH_BITPCK) {
						const uint32_t bk = mul24(r, iw.z) >> 16;
						lds_or_bits(tail32, bk * 128u + (r - bk * iw.x) * inb, gap);
					} else {
						tail[io++] = uint8_t((gap & 0x7fu) | (gap >= 128u ? 0x80u : 0u));
						if (gap >= 128u)
							tail[io++] = uint8_t(gap >> 7);
					}
					if (venc == OROCH_BITPCK) {
						const uint32_t bk = mul24(r, vw.z) >> 16;
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
	return plan;
}

// Second half, one piece of the octet at a time (all eight blocks, or blocks
// 0..3 and 4..7 of an octet too long for the image): packs the words of the
// piece's blocks into the image in LDS and copies their headers and lists
// behind them.  rel: this lane's block's place in the image.
template <typename E>
__device__ __forceinline__ void encode_octo_image(const oct_plan &plan, bool in_piece, uint32_t rel, int lane,
						  const typename E::C *slots, const uint8_t *tails,
						  const uint8_t *heads, uint8_t *img)
{
	asm volatile("" : "+v"(lane));
	const oct_lane l = make_oct_lane(lane);
	const uint32_t g = l.g, sub = l.sub;
	const bool mine = plan.ok && in_piece;
	// blocks of one width are packed together
	for (uint32_t left = oct_mask(__ballot(mine)); left;) {
		const uint32_t nb_u = read_lane(plan.nb, int(__builtin_ctz(left)) * OG);
		const uint32_t cls = oct_mask(__ballot(mine && plan.nb == nb_u));
		pack_oct_dispatch<E, FIELDS_MIN_NB, E::W>(nb_u, make_oct_class(cls), slots, rel + plan.doff, lane, img);
		left &= ~cls;
	}
	if (mine) {
		const uint8_t *head = heads + g * 16;
		if (sub * 8u < plan.doff)
			*reinterpret_cast<uint2 *>(img + rel + sub * 8) = *reinterpret_cast<const uint2 *>(head + sub * 8);
		// lists and the zero padding up to the block's 16-byte boundary
		const uint8_t *tail = tails + g * OT_BYTES;
		for (uint32_t k = sub * 8u; k < plan.size - plan.list_at; k += OG * 8u)
			*reinterpret_cast<uint2 *>(img + rel + plan.list_at + k) = *reinterpret_cast<const uint2 *>(tail + k);
	}
	wave_sync();
}

// ---- the launch -------------------------------------------------------------------
//
// A tile is one octet (eight consecutive blocks).  A workgroup is OC_WORKERS
// wavefronts that select and encode (workers), one that finds out where the
// workgroup's tiles go (the exchanger) and one that copies finished tiles to
// the stream (the writer); one workgroup per CU, all resident.  In round r the
// workers of workgroup b take the consecutive tiles (r * B + b) * OC_WORKERS ..
// + OC_WORKERS - 1, B being the number of workgroups.
//
// The three kinds of wavefront exist because a wavefront's memory operations
// complete in order: a load issued behind a store waits until the store has
// been acknowledged, several microseconds under load.  So the workers only
// load (the values, one tile ahead), the exchanger only loads (the table
// below), the writer only stores, and they talk to each other through LDS.
//
// Where a tile goes is the sum of the sizes of all tiles before it, which other
// workgroups are working out at the same time.  The sizes are exchanged through
// one small table in device memory, zero before the launch:
//
//   row[r][b]  16 bits: bit 15 set once workgroup b has the sizes of all its
//              tiles of round r, bits 14..0 their sum in units of 16 bytes;
//              stored by the writer once the workgroup's workers have all
//              posted their sizes of that round.
//
// The offset of workgroup b's tiles of round r is then the sum of all complete
// rows before r, which the exchanger keeps as it goes, plus row[r][0 .. b-1]:
// one hop from the stores of the others to its own loads, the whole row (512
// bytes for 256 workgroups) in one load per lane.
//
// A worker selects, posts its tile's size, lays out header and lists in LDS,
// asks for the next tile's values, packs the tile into its image in LDS and
// hands the image to the writer; it waits only when the writer still holds the
// image of the tile before.  Blocks the eight-block path leaves out, and the
// blocks of a tile at the array's end, go to the writer one by one from the
// one-block encoder's image.
//
// Every entry of the table is written by one agent-scope store and polled with
// agent-scope loads; flag and value travel in the same word, so no fence is
// needed.  Waits are bounded: when one runs out the abort word is set, nothing
// more is written, the stream length is left at ~0 and the host falls back to
// the separate launches.

#ifndef OC_NWORKERS
#define OC_NWORKERS 10
#endif
#ifndef OC_EU
#define OC_EU 3
#endif
constexpr int OC_WORKERS = OC_NWORKERS;
constexpr int OC_THREADS = WAVE * (OC_WORKERS + 2);
constexpr int OC_MAX_GROUPS = 256;        // workgroups = entries of a row (one load of 8 bytes per lane)
constexpr uint32_t OC_ROW_VALID = 0x8000u;
constexpr uint32_t OC_IDLE_LIMIT = 1u << 22;
constexpr int OC_DEPTH = 4;               // rounds a workgroup may have under way (a power of two)
// the image of a tile in LDS: any four blocks fit, and most octets
constexpr int OC_IMAGE_BYTES = 4 * (BLOCK * 4 + 16);
static_assert(OC_WORKERS * OCT * (BLOCK * 4 + 16) / 16 < int(OC_ROW_VALID), "a row entry holds a workgroup's round");

#ifdef OC_STATS // experiment builds: cycles per phase, summed over the workers, printed by the host
__device__ unsigned long long oc_stats[16];
#define OC_T(var) const uint64_t var = __builtin_amdgcn_s_memtime()
#define OC_ADD(i, t0, t1) do { octs[i] += uint32_t((t1) - (t0)); } while (0)
#else
#define OC_T(var) do { } while (0)
#define OC_ADD(i, t0, t1) do { } while (0)
#endif

struct oc_state
{
	uint16_t *rows;       // [rounds][OC_MAX_GROUPS]
	uint32_t *abort_flag;
};

// A piece of the stream that a worker has ready in LDS.
struct oc_piece
{
	uint32_t src;      // where it lies, in bytes from the start of the worker's LDS
	uint32_t len;      // its length, a multiple of 16
	uint32_t at;       // its place in the tile
	uint32_t round;    // the round of the tile
	uint32_t nblocks;  // with this piece go the offsets of that many blocks (starts[]), 0 for none
	uint32_t total;    // the tile's length
	uint32_t last;     // the worker's last piece of this tile
	uint32_t pad;
	uint32_t starts[OCT]; // where the tile's blocks start
};

// What the wavefronts of a workgroup tell each other.  Round r sits at [r % depth].
struct oc_mail
{
	uint32_t bytes[OC_DEPTH][OC_WORKERS];  // size of each worker's tile (the worker writes)
	uint32_t arrived[OC_DEPTH];            // workers that have written theirs << 24 | sum of sizes / 16
	uint64_t base[OC_DEPTH][OC_WORKERS];   // offset of each worker's tile (the exchanger writes)
	uint32_t published;                    // rounds whose row entry has been stored (the writer writes)
	uint32_t resolved;                     // rounds whose offset is in `base` (the exchanger writes)
	uint32_t abort;                        // the launch has been given up
	uint32_t pad;
	oc_piece piece[OC_WORKERS];            // the piece each worker has ready (the worker writes)
	uint32_t ready[OC_WORKERS];            // pieces the worker has handed over
	uint32_t done[OC_WORKERS];             // pieces the writer has copied
};

// Words in LDS that another wavefront of the workgroup writes or reads.  The
// LDS serves a wavefront's accesses in the order it issued them, so data
// written before a counter is there for whoever has seen the counter; all that
// is needed is that the compiler keeps the order too.  (Acquire and release
// forms would also wait for every global load in flight.)
__device__ __forceinline__ uint32_t oc_lds_load(const uint32_t *p)
{
	asm volatile("" ::: "memory");
	const uint32_t x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	asm volatile("" ::: "memory");
	return x;
}
__device__ __forceinline__ void oc_lds_store(uint32_t *p, uint32_t x)
{
	asm volatile("" ::: "memory");
	__hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	asm volatile("" ::: "memory");
}

// A row of the table, four entries to a lane.
__device__ __forceinline__ uint64_t oc_row_load(const oc_state &st, uint32_t round, int lane)
{
	const uint64_t *p = reinterpret_cast<const uint64_t *>(st.rows + size_t(round) * OC_MAX_GROUPS) + lane;
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The sum (in units of 16 bytes) of the first `count` entries of a row held four
// to a lane, if they are all there.
__device__ __forceinline__ bool oc_row_sum(uint64_t x, uint32_t count, int lane, uint32_t &sum)
{
	uint32_t part = 0;
	bool missing = false;
#pragma unroll
	for (int i = 0; i < 4; i++) {
		const uint32_t e = uint32_t(x >> (16 * i)) & 0xffffu;
		const bool wanted = uint32_t(4 * lane + i) < count;
		missing = missing || (wanted && !(e & OC_ROW_VALID));
		part += wanted ? e & (OC_ROW_VALID - 1u) : 0u;
	}
	if (__ballot(missing) != 0)
		return false;
	sum = wave_sum(part);
	return true;
}

// A wait has run out: the launch is given up (the abort word is set, the
// workgroup's other wavefronts are told).
__device__ __forceinline__ void oc_give_up(const oc_state &st, oc_mail *mail, int lane)
{
	if (lane == 0)
		(void) atomicCAS(st.abort_flag, 0u, 1u);
	oc_lds_store(&mail->abort, 1u);
}

// The exchanger's loop: for every round of the workgroup, once the workgroup's
// own entry is in the table, polls the table until the entries before it are
// there, and leaves the offsets of the workgroup's tiles in the mail.  Loads only.
__device__ __forceinline__ void oc_exchange(const oc_state &st, oc_mail *mail, uint32_t group, uint32_t groups,
					    uint32_t rounds, int lane)
{
	uint64_t base = 0;      // bytes of all tiles of the rounds before `rows_done`
	uint32_t rows_done = 0;
	uint32_t idle = 0;
	for (uint32_t r = 0; r < rounds; r++) {
		// the workgroup's own sizes first
		while (oc_lds_load(&mail->published) <= r) {
			if (++idle > OC_IDLE_LIMIT || oc_lds_load(&mail->abort) ||
			    ((idle & 1023u) == 0 && __hip_atomic_load(st.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
				oc_give_up(st, mail, lane);
				return;
			}
			__builtin_amdgcn_s_sleep(2);
		}
		for (;;) {
			const bool need_prev = rows_done < r; // then rows_done == r - 1
			uint64_t prev = 0;
			if (need_prev)
				prev = oc_row_load(st, r - 1u, lane);
			uint64_t cur = 0;
			if (group != 0)
				cur = oc_row_load(st, r, lane);
			asm volatile("" : "+v"(prev), "+v"(cur));
			bool ok = true;
			if (need_prev) {
				uint32_t total = 0;
				ok = oc_row_sum(prev, groups, lane, total);
				if (ok) {
					base += uint64_t(total) << 4;
					rows_done = r;
				}
			}
			uint32_t before = 0;
			ok = ok && (group == 0 || oc_row_sum(cur, group, lane, before));
			if (ok) {
				// the workgroup's tiles of a round follow one another
				const uint32_t slot = r % uint32_t(OC_DEPTH);
				const uint32_t mine = lane < OC_WORKERS ? mail->bytes[slot][lane] : 0u;
				const uint32_t upto = wave_scan<uint32_t>(mine, 0u, [](uint32_t a, uint32_t b) { return a + b; });
				if (lane < OC_WORKERS)
					mail->base[slot][lane] = base + (uint64_t(before) << 4) + (upto - mine);
				oc_lds_store(&mail->resolved, r + 1u);
				idle = 0;
				break;
			}
			if (++idle > OC_IDLE_LIMIT || oc_lds_load(&mail->abort) ||
			    ((idle & 63u) == 0 && __hip_atomic_load(st.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
				oc_give_up(st, mail, lane);
				return;
			}
			__builtin_amdgcn_s_sleep(8);
		}
	}
}

// The writer's loop.  Two duties in turn: the workgroup's entry of the table
// for the next round whose sizes are all there, and the pieces the workers have
// ready, once the round they belong to has its offsets.  Stores only.
// tiles: how many tiles the workgroup's workers have in all.
__device__ __forceinline__ void oc_write(const oc_state &st, oc_mail *mail, const uint8_t *lds, size_t worker_stride,
					 uint32_t group, uint32_t groups, uint32_t rounds, size_t ntiles,
					 size_t nblocks, uint8_t *dst, uint64_t *offsets, int lane)
{
	const size_t per_round = size_t(groups) * OC_WORKERS;
	const size_t first0 = size_t(group) * OC_WORKERS;
	uint32_t tiles_left = 0;
	for (uint32_t r = 0; r < rounds; r++) {
		const size_t tile0 = first0 + size_t(r) * per_round;
		tiles_left += ntiles - tile0 < size_t(OC_WORKERS) ? uint32_t(ntiles - tile0) : uint32_t(OC_WORKERS);
	}
	uint32_t pr = 0; // rounds published
	uint32_t idle = 0;
#ifdef OC_FAKE_OFFSETS // timing experiments: no exchange, every tile at a place of its own
	pr = rounds;
#endif
	// the workgroup's entry of the table for the next round, if its sizes are all there
	auto publish = [&]() -> bool {
		if (pr >= rounds)
			return false;
		const uint32_t slot = pr % uint32_t(OC_DEPTH);
		const size_t tile0 = first0 + size_t(pr) * per_round;
		const uint32_t nw = ntiles - tile0 < size_t(OC_WORKERS) ? uint32_t(ntiles - tile0) : uint32_t(OC_WORKERS);
		const uint32_t a = oc_lds_load(&mail->arrived[slot]);
		if ((a >> 24) != nw)
			return false;
		if (lane == 0) {
			__hip_atomic_store(&st.rows[size_t(pr) * OC_MAX_GROUPS + group], uint16_t(OC_ROW_VALID | (a & 0xffffffu)),
					   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			// (the slot's next use is OC_DEPTH rounds away)
			mail->arrived[slot] = 0;
		}
		pr++;
		oc_lds_store(&mail->published, pr);
		return true;
	};
	while (tiles_left != 0 || pr < rounds) {
		bool busy = publish();
#ifdef OC_FAKE_OFFSETS
		const uint32_t have = ~0u;
#else
		const uint32_t have = oc_lds_load(&mail->resolved);
#endif
		// which workers have a piece ready whose round has its offsets: lane w looks at worker w
		uint32_t seq = 0;
		bool ready = false;
		if (lane < OC_WORKERS) {
			seq = mail->done[lane];
			ready = mail->ready[lane] != seq && mail->piece[lane].round < have;
		}
		asm volatile("" ::: "memory");
		for (uint64_t todo = __ballot(ready); todo; todo &= todo - 1) {
			const int w = __builtin_ctzll(todo);
			// (others wait for the entry of the table: it goes first whenever it can)
			busy = publish() || busy;
			const oc_piece *pc = &mail->piece[w];
			// the piece's fields, one to a lane
			const uint32_t field = reinterpret_cast<const uint32_t *>(pc)[lane & 15];
			const uint32_t src_at = read_lane(field, 0), len = read_lane(field, 1), at = read_lane(field, 2);
			const uint32_t round = read_lane(field, 3), nblk = read_lane(field, 4), total = read_lane(field, 5);
			const uint32_t last = read_lane(field, 6), start = dpp_mov<0x108>(0u, field); // lanes 0..7: starts[lane]
			const uint8_t *src = lds + size_t(w) * worker_stride + src_at;
#ifdef OC_FAKE_OFFSETS
			const uint64_t base = uint64_t(first0 + size_t(round) * per_round + size_t(w)) * (OCT * (BLOCK * 4 + 16));
#else
			const uint64_t base = uniform(mail->base[round % uint32_t(OC_DEPTH)][w]);
#endif
			uint8_t *out = dst + base + at;
			// (a piece is at most OC_IMAGE_BYTES long: all of it is asked for at
			// once, lanes past its end read its last 16 bytes again)
			constexpr int PASSES = (OC_IMAGE_BYTES + WAVE * 16 - 1) / (WAVE * 16);
			uint4 x[PASSES];
#pragma unroll
			for (int k = 0; k < PASSES; k++) {
				const uint32_t o = uint32_t(lane) * 16u + uint32_t(k) * WAVE * 16u;
				x[k] = *reinterpret_cast<const uint4 *>(src + (o < len ? o : len - 16u));
			}
#pragma unroll
			for (int k = 0; k < PASSES; k++) {
				const uint32_t o = uint32_t(lane) * 16u + uint32_t(k) * WAVE * 16u;
				if (o < len)
					*reinterpret_cast<uint4 *>(out + o) = x[k];
			}
			if (nblk) {
				const size_t tile = first0 + size_t(round) * per_round + size_t(w);
				if (uint32_t(lane) < nblk)
					offsets[tile * OCT + size_t(lane)] = base + start;
				if (tile == ntiles - 1 && lane == 0)
					offsets[nblocks] = base + total;
			}
			tiles_left -= last;
			// (the piece's bytes are in registers by now: its place may be written again)
			oc_lds_store(&mail->done[w], read_lane(seq, w) + 1u);
			busy = true;
		}
		if (busy) {
			idle = 0;
		} else {
			if (++idle > OC_IDLE_LIMIT || oc_lds_load(&mail->abort) ||
			    ((idle & 1023u) == 0 && __hip_atomic_load(st.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
				oc_give_up(st, mail, lane);
				return;
			}
			__builtin_amdgcn_s_sleep(1);
		}
	}
}

// LDS per worker: a work area that holds the selection's tables during
// selection and the value slots (or a one-block image) during encoding, the
// staging of headers and lists, and the image of the tile.
template <typename E>
constexpr int oc_work_bytes()
{
	constexpr int sel = (OCT * OH_DWORDS + OCT * ON_DWORDS + OCT * OP_DWORDS) * 4;
	constexpr int enc = int(sizeof(typename E::C)) * OCT * OCT_SLOT_STRIDE;
	constexpr int img = image_bytes<E>();
	constexpr int m = sel > enc ? (sel > img ? sel : img) : (enc > img ? enc : img);
	return (m + 15) / 16 * 16;
}
template <typename E>
struct oc_worker_lds
{
	alignas(16) uint8_t work[oc_work_bytes<E>()];
	alignas(16) uint8_t tails[OCT * OT_BYTES];
	alignas(16) uint8_t heads[OCT * 16];
	alignas(16) uint8_t image[OC_IMAGE_BYTES];
};

template <typename E>
__global__ __launch_bounds__(OC_THREADS) __attribute__((amdgpu_waves_per_eu(OC_EU, OC_EU))) void k_fused_octo(
	const typename E::S *__restrict__ src, size_t n, size_t nblocks, size_t ntiles,
	uint8_t *__restrict__ dst, uint64_t *__restrict__ offsets, oc_state st)
{
	using C = typename E::C;
	__shared__ oc_worker_lds<E> wl[OC_WORKERS];
	__shared__ oc_mail mail;
	__shared__ uint4 wt[65];
	const int wave = uniform(int(threadIdx.x / WAVE)), lane = lane_id();
	if (threadIdx.x < OC_DEPTH)
		mail.arrived[threadIdx.x] = 0;
	if (threadIdx.x < OC_WORKERS) {
		mail.ready[threadIdx.x] = 0;
		mail.done[threadIdx.x] = 0;
	}
	if (threadIdx.x == 0) {
		mail.published = 0;
		mail.resolved = 0;
		mail.abort = 0;
	}
	load_width_table(wt); // (ends with the workgroup's only barrier)
	const uint32_t group = blockIdx.x, groups = gridDim.x;
	const size_t per_round = size_t(groups) * OC_WORKERS;
	// rounds of the workgroup = rounds of its first worker
	const size_t first0 = size_t(group) * OC_WORKERS;
	const uint32_t wg_rounds = first0 < ntiles ? uint32_t((ntiles - first0 + per_round - 1) / per_round) : 0u;
	if (wave == OC_WORKERS) {
		// (little to do and others wait for it: ahead of the workers when it has something)
		__builtin_amdgcn_s_setprio(3);
		oc_exchange(st, &mail, group, groups, wg_rounds, lane);
		return;
	}
	if (wave == OC_WORKERS + 1) {
		__builtin_amdgcn_s_setprio(3);
		oc_write(st, &mail, reinterpret_cast<const uint8_t *>(&wl[0]), sizeof(oc_worker_lds<E>), group, groups,
			 wg_rounds, ntiles, nblocks, dst, offsets, lane);
		return;
	}
	const lane_consts lc = make_lane_consts(wt, lane);
	oc_worker_lds<E> &my = wl[wave];
	// selection's view of the work area
	uint32_t *hq = reinterpret_cast<uint32_t *>(my.work);
	uint32_t *no = hq + OCT * OH_DWORDS;
	uint32_t *pw = no + OCT * ON_DWORDS;
	// the encoder's view
	C *slots = reinterpret_cast<C *>(my.work);
	uint32_t *image1 = reinterpret_cast<uint32_t *>(my.work); // the one-block encoder's
	oc_piece *piece = &mail.piece[wave];
	uint32_t handed = 0; // pieces handed to the writer

	const uint32_t g = uint32_t(lane) >> 3, sub = uint32_t(lane) & 7u;
	const size_t first = first0 + size_t(wave);
	const uint32_t rounds = first < ntiles ? uint32_t((ntiles - first + per_round - 1) / per_round) : 0u;

	auto is_whole = [&](size_t t) -> bool {
		return t < ntiles && t * OCT + OCT <= nblocks && (t * OCT + OCT) * BLOCK <= n;
	};
	// waits until the writer has copied every piece handed over; false if the launch was given up
	auto writer_done = [&]() -> bool {
		uint32_t idle = 0;
		while (oc_lds_load(&mail.done[wave]) != handed) {
			if (oc_lds_load(&mail.abort))
				return false;
			if (++idle > OC_IDLE_LIMIT) {
				oc_give_up(st, &mail, lane);
				return false;
			}
			__builtin_amdgcn_s_sleep(1);
		}
		return true;
	};
	// hands a piece to the writer (the piece's bytes are in LDS by now)
	auto hand_over = [&](uint32_t src_at, uint32_t len, uint32_t at, uint32_t round, uint32_t nblk, uint32_t total,
			     uint32_t last, uint32_t start) {
		if (lane == 0) {
			piece->src = src_at;
			piece->len = len;
			piece->at = at;
			piece->round = round;
			piece->nblocks = nblk;
			piece->total = total;
			piece->last = last;
		}
		if (nblk && (lane & (OG - 1)) == 0)
			piece->starts[lane >> 3] = start;
		handed++;
		wave_sync();
		oc_lds_store(&mail.ready[wave], handed);
	};
	// The values of a tile are asked for while the tile before it is still being
	// packed.  (Loaded without a condition: an array that is only written under
	// one is kept in scratch memory; a tile that is not whole reads the first
	// tile's values, which every array that gets here has, and ignores them.)
	auto values_at = [&](size_t t) {
		return src + ((is_whole(t) ? t * OCT : size_t(0)) + size_t(g)) * BLOCK + size_t(sub) * VPO;
	};
	C d[VPO];
	load32<E>(values_at(first), d);
#ifdef OC_STATS
	uint32_t octs[8] = {};
#endif
	for (uint32_t r = 0; r < rounds; r++) {
		const size_t tile = first + size_t(r) * per_round;
		const size_t blk0 = tile * OCT;
		const bool whole = is_whole(tile);
		quad_selection sel;
		C mn = 0;
		uint32_t msel = 0;
		uint32_t size = 0; // of block g (whole tiles) or of block `lane / 8` in its first lane (others)
		OC_T(t0);
#ifdef OC_STATS
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
		OC_T(t1);
		OC_ADD(0, t0, t1);
		if (whole) {
			select_octo<E>(d, lane, hq, no, pw, wt, sel, mn, msel);
			wave_sync();
			size = stream_size(sel);
		} else {
			for (int b = 0; b < OCT && blk0 + size_t(b) < nblocks; b++) {
				const size_t blk = blk0 + size_t(b);
				const int nv = block_count(n, blk);
				C w[VPL];
				load_block<E>(src, blk * BLOCK, nv, lane, w);
				selection<C> s1;
				select_any<E>(w, nv, lane, hq, wt, lc, s1);
				size = lane == b * OG ? stream_size(s1) : size;
				wave_sync();
			}
		}
		// where the blocks start inside the tile, and the tile's size
		uint32_t rel, total;
		{
			const uint32_t mine = sub == 0 ? size : 0u;
			const uint32_t upto = wave_scan<uint32_t>(mine, 0u, [](uint32_t a, uint32_t b) { return a + b; });
			// (whole tiles: in every lane of a block; others: in the block's first lane)
			rel = upto - (whole ? size : mine);
			total = read_lane(upto, WAVE - 1);
		}
		// this tile's size goes to the mail
#ifndef OC_FAKE_OFFSETS
		if (lane == 0) {
			const uint32_t slot = r % uint32_t(OC_DEPTH);
			mail.bytes[slot][wave] = total;
			asm volatile("" ::: "memory");
			(void) __hip_atomic_fetch_add(&mail.arrived[slot], (1u << 24) | (total >> 4), __ATOMIC_RELAXED,
						      __HIP_MEMORY_SCOPE_WORKGROUP);
		}
#endif
		OC_T(t2);
		OC_ADD(1, t1, t2);
		// what can be done while the writer may still hold the image
		oct_plan plan;
		plan.fast = 0;
		plan.ok = false;
		if (whole)
			plan = encode_octo_prepare<E>(d, mn, sel, msel, lane, slots, my.tails, my.heads, wt);
		// the next tile's values (the deltas of this one are in the slots by now)
		load32<E>(values_at(tile + per_round), d);
		OC_T(t3);
		OC_ADD(2, t2, t3);
		const uint32_t nblk = nblocks - blk0 < size_t(OCT) ? uint32_t(nblocks - blk0) : uint32_t(OCT);
		const uint32_t all = (1u << OCT) - 1u;
		bool first_piece = true;
		if (plan.fast) {
			// one piece if the tile fits the image, else its halves (any four blocks fit)
			const uint32_t mid = read_lane(rel, 4 * OG);
			const bool halves = total > uint32_t(OC_IMAGE_BYTES);
			for (uint32_t p = 0; p < (halves ? 2u : 1u); p++) {
				const bool in_piece = !halves || (g >> 2) == p;
				const uint32_t at = halves && p ? mid : 0u;
				const uint32_t len = halves ? (p ? total - mid : mid) : total;
				OC_T(t4);
				if (!writer_done())
					return;
				OC_T(t5);
				OC_ADD(3, t4, t5);
				encode_octo_image<E>(plan, in_piece, rel - at, lane, slots, my.tails, my.heads, my.image);
				OC_T(t6);
				OC_ADD(4, t5, t6);
				const bool last = plan.fast == all && p + 1u == (halves ? 2u : 1u);
				hand_over(uint32_t(my.image - my.work), len, at, r, first_piece ? nblk : 0u, total, last ? 1u : 0u,
					  rel);
				first_piece = false;
			}
		}
		if (plan.fast != all) {
			// the blocks the eight-block path leaves out, one by one (the pieces
			// above, if any, left holes for them)
			for (int b = 0; b < int(nblk); b++) {
				if ((plan.fast >> b) & 1u)
					continue;
				const size_t blk = blk0 + size_t(b);
				const int nv = block_count(n, blk);
				C w[VPL];
				selection<C> s1;
				load_block<E>(src, blk * BLOCK, nv, lane, w);
				if (whole) {
					const int at = b * OG;
					s1.origin = C(read_lane(sel.origin, at));
					s1.dataspace = read_lane(sel.dataspace, at);
					s1.metaspace = read_lane(sel.metaspace, at);
					s1.noutliers = read_lane(sel.noutliers, at);
					s1.encoding = read_lane(sel.encoding, at);
					s1.nbits = read_lane(sel.nbits, at);
					s1.ienc = read_lane(sel.ienc, at);
					s1.inbits = read_lane(sel.inbits, at);
					s1.venc = read_lane(sel.venc, at);
					s1.vnbits = read_lane(sel.vnbits, at);
				}
				// (the writer may still be reading the image of the block before)
				if (!writer_done())
					return;
				if (!whole) {
					select_any<E>(w, nv, lane, hq, wt, lc, s1);
					wave_sync();
				}
				const uint32_t at = read_lane(rel, b * OG);
				const uint32_t len = nv == BLOCK ? encode_block<E, true>(w, nv, lane, s1, wt, image1)
								 : encode_block<E, false>(w, nv, lane, s1, wt, image1);
				wave_sync();
				const bool last = (plan.fast >> (b + 1)) == (all >> (b + 1)) || b + 1 == int(nblk);
				hand_over(0u, (len + 15u) & ~15u, at, r, first_piece ? nblk : 0u, total, last ? 1u : 0u, rel);
				first_piece = false;
			}
			// (the work area is the next tile's, the writer must be through with it)
			if (!writer_done())
				return;
		}
		OC_T(t7);
		OC_ADD(5, t0, t7);
	}
#ifdef OC_STATS
	if (lane == 0) {
		for (int i = 0; i < 8; i++)
			atomicAdd(&oc_stats[i], (unsigned long long) octs[i]);
		atomicAdd(&oc_stats[8], (unsigned long long) rounds);
	}
#endif
}

// Selection alone, eight blocks per wavefront: the descriptors of the full
// octets of an array (the rest is left to the caller).
#ifndef SELECT_OCTO_WAVES
#define SELECT_OCTO_WAVES 4
#endif
template <typename E>
__global__ __launch_bounds__(WAVE *WAVES_PER_WG) __attribute__((amdgpu_waves_per_eu(SELECT_OCTO_WAVES, 8))) void k_select_octo(const typename E::S *__restrict__ src,
								      size_t noctets,
								      oroch_block_desc *__restrict__ desc,
								      uint32_t *__restrict__ sizes)
{
	using C = typename E::C;
	__shared__ __attribute__((aligned(16))) uint32_t hqs[WAVES_PER_WG][OCT * OH_DWORDS];
	__shared__ __attribute__((aligned(16))) uint32_t nos[WAVES_PER_WG][OCT * ON_DWORDS];
	__shared__ __attribute__((aligned(16))) uint32_t pws[WAVES_PER_WG][OCT * OP_DWORDS];
	__shared__ uint4 wt[65];
	const int wave = uniform(int(threadIdx.x / WAVE)), lane = lane_id();
	load_width_table(wt);
	const size_t stride = size_t(gridDim.x) * WAVES_PER_WG;
	for (size_t q = size_t(blockIdx.x) * WAVES_PER_WG + wave; q < noctets; q += stride) {
		const uint32_t g = uint32_t(lane) >> 3, sub = uint32_t(lane) & 7u;
		const size_t blk = q * OCT + size_t(g);
		C d[VPO];
		load32<E>(src + blk * BLOCK + size_t(sub) * VPO, d);
		quad_selection sel;
		C mn;
		uint32_t msel;
		select_octo<E>(d, lane, hqs[wave], nos[wave], pws[wave], wt, sel, mn, msel);
		uint4 da, db;
		desc_words(sel, da, db);
		if (desc && sub < 2) {
			const uint4 half = make_uint4(sub ? db.x : da.x, sub ? db.y : da.y, sub ? db.z : da.z,
						      sub ? db.w : da.w);
			reinterpret_cast<uint4 *>(desc + blk)[sub] = half;
		}
		if (sizes && sub == 0)
			sizes[blk] = da.w;
		wave_sync();
	}
}

// Workgroups of k_fused_octo the launch may have: all must be resident (a tile
// may wait for any lower-numbered one), and a row of the table holds 256.
template <typename E>
static int oc_resident_workgroups()
{
	static int cached = 0;
	if (cached)
		return cached;
	int dev = 0, cus = 0, per_cu = 0;
	if (hipGetDevice(&dev) != hipSuccess ||
	    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
	    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fused_octo<E>, OC_THREADS, 0) != hipSuccess)
		return -1;
	if (per_cu < 1 || cus < 1)
		return -1;
	// one per CU even where a second would fit
	cached = cus < OC_MAX_GROUPS ? cus : OC_MAX_GROUPS;
	return cached;
}

} // namespace oroch_hip
