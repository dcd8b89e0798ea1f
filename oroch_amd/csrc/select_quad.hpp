// select_quad.hpp -- codec selection for four consecutive full blocks per
// wavefront, sixteen lanes per block (integer_codec.h:234-384 with
// integer_stats.h:48-53, 74-80).  Same rule and same candidate order as
// select_full; what changes is where the work sits: everything that is uniform
// per block lives in the vector registers of the block's 16-lane group, so the
// four blocks advance together and nothing is left for the scalar unit but the
// loop control.  Reductions and scans are DPP row operations (a row is a
// group).  Each lane looks after two candidate widths, sub + 1 and sub + 17.
//
// Included by group.hip after select_full and encode_quad.hpp; 8-, 16- and
// 32-bit element types.
#pragma once

namespace oroch_hip {

// Histogram bins are kept in HQ_COPIES copies each, one per lane number modulo
// HQ_COPIES, so that fewer lanes of a group add to the same word in one
// instruction: with the usual narrow spread of bit lengths most of a group's
// sixteen lanes hit two or three bins, and words that several lanes add to are
// served one lane after the other.
#ifndef HQ_COPIES
#define HQ_COPIES 4
#endif
constexpr int HQ_BINS = 35;     // "zero or minus one", zero, and bit lengths 32 down to 1
// per group: HQ_BINS x HQ_COPIES words, padded so that the four groups' tables sit
// HQ_COPIES banks apart (16 copies: half the banks)
constexpr int HQ_DWORDS = HQ_COPIES == 4 ? 196 : HQ_COPIES == 8 ? 296 : 560;
static_assert(HQ_COPIES == 4 || HQ_COPIES == 8 || HQ_COPIES == 16, "copies per bin");
static_assert(HQ_DWORDS >= HQ_BINS * HQ_COPIES && HQ_DWORDS % 4 == 0, "room for every bin");
constexpr int NO_DWORDS = 64;   // per group: exception counts by width, zero above the widest
constexpr int PW_ENTRIES = 34;  // per group: per-width candidate parameters

template <typename V, typename Op>
__device__ __forceinline__ V row_allreduce(V x, V ident, Op op)
{
	x = op(x, dpp_mov<0xb1>(ident, x));
	x = op(x, dpp_mov<0x4e>(ident, x));
	x = op(x, dpp_mov<0x124>(ident, x));
	x = op(x, dpp_mov<0x128>(ident, x));
	return x;
}
__device__ __forceinline__ uint32_t row_sum(uint32_t x)
{
	return row_allreduce(x, 0u, [](uint32_t a, uint32_t b) { return a + b; });
}
__device__ __forceinline__ uint32_t row_max(uint32_t x)
{
	return row_allreduce(x, 0u, [](uint32_t a, uint32_t b) { return a > b ? a : b; });
}
__device__ __forceinline__ uint32_t row_min(uint32_t x)
{
	return row_allreduce(x, ~0u, [](uint32_t a, uint32_t b) { return a < b ? a : b; });
}

// v_ffbh_u32: leading zeros of x, -1 for x == 0 (the builtin's result is
// undefined there, the instruction's is not).
__device__ __forceinline__ int leading_zeros(uint32_t x)
{
	int r;
	asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x));
	return r;
}
// v_ffbh_i32: how many leading bits of x equal its sign bit, -1 for 0 and -1.
__device__ __forceinline__ int leading_sign_bits(int32_t x)
{
	int r;
	asm("v_ffbh_i32 %0, %1" : "=v"(r) : "v"(x));
	return r;
}
// Multiplies of small numbers: the 24-bit multiply runs at full rate, the
// 32-bit ones at a quarter of it.
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b)
{
	return uint32_t(__umul24(a, b));
}
// 16 * ceil(count / capacity): bitpck.h:61-71 for count + capacity - 1 < 512,
// w = {capacity, ., ceil(2^16 / capacity), .}
__device__ __forceinline__ uint32_t pck_bytes(uint32_t count, const uint4 &w)
{
	return (mul24(count + w.x - 1u, w.z) >> 16) << 4;
}

// The 16 bits of a ballot that belong to this lane's group.
__device__ __forceinline__ uint32_t group_bits(uint64_t ballot, int g)
{
	const uint32_t half = (g & 2) ? uint32_t(ballot >> 32) : uint32_t(ballot);
	return (half >> ((g & 1) * 16)) & 0xffffu;
}

// What a lane needs to know about its two widths, computed once per kernel.
struct quad_lane_consts
{
	uint32_t wA, wB;     // varint bytes of a value of bit length sub + 1, sub + 17
	uint32_t pckA, pckB; // bitpck space of 256 values at those widths
};

__device__ __forceinline__ quad_lane_consts make_quad_lane_consts(const uint4 *wt, int lane)
{
	const uint32_t sub = uint32_t(lane) & 15u;
	quad_lane_consts c;
	c.wA = vsp_bits(sub + 1u);
	c.wB = vsp_bits(sub + 17u);
	c.pckA = wt[sub + 1u].w;
	c.pckB = wt[sub + 17u].w;
	return c;
}

// Group-uniform outcome of selection, one copy per lane.
struct quad_selection
{
	uint32_t encoding, nbits, metaspace, dataspace, noutliers, ienc, inbits, venc, vnbits, origin;
};

// hq: QUAD * HQ_DWORDS, no: QUAD * NO_DWORDS, pw: QUAD * PW_ENTRIES uint4 of LDS
// owned by this wavefront.
template <typename E>
__device__ __forceinline__ void select_quad(const typename E::C (&v)[VPQ], int lane, uint32_t *hq,
					    uint32_t *no, uint4 *pw, const uint4 *wt, const quad_lane_consts &lc,
					    quad_selection &sel)
{
	using C = typename E::C;
	static_assert(sizeof(C) == 4, "element types up to 32 bits");
	asm volatile("" : "+v"(lane)); // see decode_quad_fast
	static_assert(GROUP == 16, "lane arithmetic below");
	const int g = int(uint32_t(lane) >> 4), sub = int(uint32_t(lane) & 15u);
	const uint32_t bA = uint32_t(sub) + 1u, bB = uint32_t(sub) + 17u;

	sel.encoding = OROCH_NORMAL;
	sel.nbits = 0;
	sel.metaspace = 1;
	sel.dataspace = 0;
	sel.noutliers = 0;
	sel.ienc = sel.venc = OROCH_NORMAL;
	sel.inbits = sel.vnbits = 0;
	sel.origin = 0;

	// integer_stats.h:88-95: the minimum (signed compare for int32, else on
	// order-preserving unsigned keys)
	C mn;
	if constexpr (E::sgn && E::W == 32) {
		int32_t smin = int32_t(v[0]);
#pragma unroll
		for (int i = 1; i < VPQ; i++)
			smin = int32_t(v[i]) < smin ? int32_t(v[i]) : smin;
		mn = C(row_min(uint32_t(smin) ^ 0x80000000u) ^ 0x80000000u);
	} else {
		C kmin = v[0] ^ E::FLIP;
#pragma unroll
		for (int i = 1; i < VPQ; i++) {
			const C k = v[i] ^ E::FLIP;
			kmin = k < kmin ? k : kmin;
		}
		mn = row_min(kmin) ^ E::FLIP;
	}
	const uint32_t vo = vsp(zz<E>(mn));

	// Bit-length histogram of v - min, integer_stats.h:97-102.  Bins are indexed
	// by leading zeros + 1 (v_ffbh_u32 gives -1 for zero, so bin 0 is "value
	// zero" and bin c + 1 is bit length 32 - c) and are HQ_COPIES words each (one
	// per sub modulo HQ_COPIES, to spread same-address atomics).
	constexpr int NC = HQ_COPIES;
	uint32_t *table = hq + mul24(uint32_t(g), HQ_DWORDS) + NC;
#pragma unroll
	for (int k = 0; k < (HQ_BINS * NC + GROUP * 4 - 1) / (GROUP * 4); k++)
		if ((k * GROUP + sub) * 4 < HQ_DWORDS)
			*reinterpret_cast<uint4 *>(&table[(k * GROUP + sub) * 4 - NC]) = make_uint4(0, 0, 0, 0);
	uint32_t *nog = no + g * NO_DWORDS;
	nog[32 + sub] = 0;
	nog[48 + sub] = 0;
	wave_sync();
	uint32_t *mine = table + NC + (sub & (NC - 1));
	// leading-zero counts of the deltas, four to a register (value 4j + r in
	// byte j of lz[r]; 255 for a zero delta)
	int cds[VPQ];
	// The zigzag images need no histogram: what selection wants of them is the
	// sum of their varint lengths and the longest of them.  A value whose
	// zigzag image is L bits long takes ceil(L / 7) = (37 L + 219) >> 8 bytes
	// (L in 1..32; zero takes one byte like L = 1), and L is ZL minus the
	// count of redundant leading bits, clamped for the "none at all" answer.
	constexpr uint32_t ZL = E::sgn ? 33u : 32u, ZCAP = ZL - 1u;
	uint32_t zbytes = 0, zlead = ZCAP;
#pragma unroll
	for (int i = 0; i < VPQ; i++) {
		const int cd = leading_zeros(uint32_t((v[i] - mn) & E::MASK));
		int cz;
		if constexpr (E::sgn)
			cz = leading_sign_bits(int32_t(v[i] << (32 - E::W)) >> (32 - E::W));
		else
			cz = leading_zeros(uint32_t(v[i]));
		atomicAdd(&mine[cd * NC], 1u);
		const uint32_t czc = uint32_t(cz) < ZCAP ? uint32_t(cz) : ZCAP; // -1 reads as huge
		zbytes += (37u * ZL + 219u - mul24(czc, 37u)) >> 8;
		zlead = czc < zlead ? czc : zlead;
		cds[i] = cd;
	}
	uint32_t lz[4];
#pragma unroll
	for (int r = 0; r < 4; r++) {
		// low bytes of four counts into one register
		const uint32_t lo = __builtin_amdgcn_perm(uint32_t(cds[r + 4]), uint32_t(cds[r]), 0x0c0c0400u);
		const uint32_t hi = __builtin_amdgcn_perm(uint32_t(cds[r + 12]), uint32_t(cds[r + 8]), 0x0c0c0400u);
		lz[r] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
	}
	wave_sync();
	// width b <-> bin 33 - b
	auto bin_sum = [&](const uint32_t *at) {
		uint32_t total = 0;
#pragma unroll
		for (int k = 0; k < NC; k += 4) {
			const uint4 q = *reinterpret_cast<const uint4 *>(at + k);
			total += q.x + q.y + q.z + q.w;
		}
		return total;
	};
	const uint32_t sa = bin_sum(&table[(33u - bA) * NC]);
	const uint32_t sb = bin_sum(&table[(33u - bB) * NC]);
	uint32_t s0 = bin_sum(&table[0]);
	const uint32_t dA = sa, dB = sb, zero_d = s0;
	const bool constant = zero_d == uint32_t(BLOCK); // integer_codec.h:255-264, applied at the end

	// the two varint footprints, integer_codec.h:483-489, from the histograms
	const uint32_t wA = lc.wA, wB = lc.wB;
	const uint32_t t = row_sum(mul24(dA, wA) + mul24(dB, wB) + (zbytes << 16));
	const uint32_t vfs = (t & 0xffffu) + zero_d, vis = t >> 16;
	// widest delta and widest zigzag image (at least one bit, bitpck.h:54-57)
	const uint32_t nbmax = row_max(dB ? bB : (dA ? bA : 0u));
	const uint32_t nbz = ZL - row_min(zlead);

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
	// values longer than b bits, for this lane's two widths
	const uint32_t pA = row_scan_sum(dA), pB = row_scan_sum(dB);
	const uint32_t totB = lane_gather(pB, g * GROUP + GROUP - 1);
	const uint32_t noutA = uint32_t(BLOCK) - zero_d - pA, noutB = totB - pB;
	nog[bA - 1] = noutA; // entry b - 1 holds the count for width b
	nog[bB - 1] = noutB;
	wave_sync();
	// varint bytes of the high parts: each value of bit length k > b costs
	// ceil((k - b) / 7), i.e. the sum over j >= 0 of nout(b + 7j) (:308-312)
	const uint32_t valvarA = noutA + nog[bA + 6] + nog[bA + 13] + nog[bA + 20] + nog[bA + 27];
	const uint32_t basic_meta = 1 + vo;
	uint4 *pwg = pw + mul24(uint32_t(g), PW_ENTRIES);
	bool candA, candB;
	uint32_t estA, estB, lowA = 0, lowB = 0; // lowX: what the width needs at the very least
	{
		const bool active = bA < nbmax && dA != 0;
		const uint32_t xmeta = 2 + (noutA < 128u ? 1u : 2u);
		const uint4 cm = wt[active ? nbmax - bA : 1u];
		const uint32_t valpck = pck_bytes(noutA, cm);
		const uint32_t venc = valpck < valvarA ? OROCH_BITPCK : OROCH_VARINT;
		const uint32_t vdata = valpck < valvarA ? valpck : valvarA;
		const uint32_t indmin = noutA > 128u ? 32u : 16u; // min(pck(nout, 1), 256), nout in 1..255
		const uint32_t fixed = basic_meta + xmeta + lc.pckA + vdata;
		const uint32_t est = fixed + indmin;
		candA = active;
		estA = est;
		lowA = fixed + (noutA < indmin ? noutA : indmin); // a varint list of one-byte gaps
		pwg[bA] = make_uint4(est, fixed, noutA | (venc << 16) | (xmeta << 24), 0);
	}
	candB = false;
	estB = 0;
	// widths 17..32 only where some block of the quad has a candidate there
	if (__ballot(bB < nbmax && dB != 0) != 0) {
		const bool active = bB < nbmax && dB != 0;
		const uint32_t valvarB = noutB + nog[bB + 6] + nog[bB + 13] + nog[bB + 20] + nog[bB + 27];
		const uint32_t xmeta = 2 + (noutB < 128u ? 1u : 2u);
		const uint4 cm = wt[active ? nbmax - bB : 1u];
		const uint32_t valpck = pck_bytes(noutB, cm);
		const uint32_t venc = valpck < valvarB ? OROCH_BITPCK : OROCH_VARINT;
		const uint32_t vdata = valpck < valvarB ? valpck : valvarB;
		const uint32_t indmin = noutB > 128u ? 32u : 16u;
		const uint32_t fixed = basic_meta + xmeta + lc.pckB + vdata;
		estB = fixed + indmin;
		lowB = fixed + (noutB < indmin ? noutB : indmin);
		candB = active;
		pwg[bB] = make_uint4(estB, fixed, noutB | (venc << 16) | (xmeta << 24), 0);
	}
	candA = candA && !constant;
	candB = candB && !constant;
	wave_sync();

	// One candidate width of each block, all four blocks in step: the index
	// list of width bL (integer_codec.h:338-351) and what the block then takes.
	auto evaluate = [&](uint32_t bL, const uint4 &par, uint32_t &required, uint32_t &how) {
		const uint32_t fixedL = par.y, noutL = par.z & 0xffffu;
		// My exceptions at this width, bit i for value i: bit length above bL,
		// i.e. fewer than 32 - bL leading zeros (a zero delta reads as 255).
		const uint32_t thr = __builtin_amdgcn_perm(32u - bL, 32u - bL, 0u); // in all four bytes
		// bit 7 of a byte comes out set iff that byte of lz is >= 32 - bL: not
		// an exception.  Flags land at bit 8j + r for value 4j + r, then move
		// together into 16 bits.
		const uint32_t c = ((((lz[0] | 0x80808080u) - thr) & 0x80808080u) >> 7) |
				   ((((lz[1] | 0x80808080u) - thr) & 0x80808080u) >> 6) |
				   ((((lz[2] | 0x80808080u) - thr) & 0x80808080u) >> 5) |
				   ((((lz[3] | 0x80808080u) - thr) & 0x80808080u) >> 4);
		const uint32_t ne = (c & 0xfu) | ((c >> 4) & 0xf0u) | ((c >> 8) & 0xf00u) | ((c >> 12) & 0xf000u);
		const uint32_t m = ~ne & 0xffffu; // my exceptions, bit i for value i
		// Index gaps, :338-351.  The gap in front of my first exception reaches
		// back to the last exception of the lanes before me; the gaps between
		// my own exceptions are runs of zeros inside m.
		// (positions count from 1 here, 0 = none, so that the scan's identity is 0)
		const uint32_t hb = 31u - uint32_t(__clz(m | 1u));
		const uint32_t lastpos1 = m ? uint32_t(sub * VPQ) + hb + 1u : 0u;
		uint32_t prev1 = dpp_mov<0x111>(0u, row_scan_max(lastpos1));
		// Keeps the lane shift an instruction of its own: folded into the
		// subtraction below (v_subrev_u32 with a DPP operand) the result was
		// wrong on the device.
		asm volatile("" : "+v"(prev1));
		const uint32_t fb = uint32_t(__builtin_ctz(m | 0x10000u));
		const uint32_t gfirst = m ? uint32_t(sub * VPQ) + fb - prev1 : 0u;
		// zeros strictly between my lowest and highest exception
		const uint32_t z1 = ~m & ((2u << hb) - 1u) & ~((1u << fb) - 1u) & 0xffffu;
		const uint32_t z2 = z1 & (z1 >> 1), z4 = z2 & (z2 >> 2), z8 = z4 & (z4 >> 4);
		const uint32_t inbits = (z1 < 1u ? z1 : 1u) + (z2 < 1u ? z2 : 1u) + (z4 < 1u ? z4 : 1u) + (z8 < 1u ? z8 : 1u);
		const uint32_t fbits = bitlen(gfirst);
		uint32_t indnbits = row_max(fbits > inbits ? fbits : inbits); // bit length of the longest gap
		// 256 positions hold at most one gap of 128 or more
		uint32_t indvar = noutL + (indnbits == 8u ? 1u : 0u);
		indnbits = indnbits ? indnbits : 1u;
		const uint4 ci = wt[indnbits];
		const uint32_t indpck = pck_bytes(noutL, ci);
		const uint32_t ienc = indpck < indvar ? OROCH_BITPCK : OROCH_VARINT;
		const uint32_t idata = indpck < indvar ? indpck : indvar;
		required = fixedL + idata;
		how = bL | (ienc == OROCH_BITPCK ? indnbits << 8 : 0u) | (ienc << 16);
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
	candA = candA && estA < best;
	candB = candB && estB < best;
	bool walk;
	{
		const uint32_t keyA = candA ? (lowA << 6) | (bA - 1u) : ~0u;
		const uint32_t keyB = candB ? (lowB << 6) | (bB - 1u) : ~0u;
		const uint32_t first = row_min(keyA < keyB ? keyA : keyB);
		const uint32_t restA = keyA == first ? ~0u : keyA, restB = keyB == first ? ~0u : keyB;
		const uint32_t second = row_min(restA < restB ? restA : restB);
		const bool any = first != ~0u;
		const uint32_t bL = any ? (first & 63u) + 1u : 1u;
		const uint4 par = pwg[bL];
		uint32_t required, how;
		evaluate(bL, par, required, how);
		const uint32_t est = par.x;
		const uint32_t top = required > est ? required : est;
		const uint32_t bound = best < (second >> 6) ? best : (second >> 6); // second >> 6 is huge when there is none
		const bool settled = top < bound;
		if (any && settled) {
			best = required;
			got = how;
			gotpar = par.z;
		}
		walk = any && !settled;
	}
	if (__ballot(walk) != 0) {
		// Replay of the reference's ascending loop for the groups that need it
		// (the others idle through): each group takes its narrowest remaining
		// candidate.  `best` only shrinks, so a width whose estimate is not
		// below it fails the gate now and ever after: each round a group moves
		// straight on to its narrowest remaining width that passes.
		candA = candA && walk;
		candB = candB && walk;
		for (;;) {
			candA = candA && estA < best;
			candB = candB && estB < best;
			const uint64_t ballA = __ballot(candA), ballB = __ballot(candB);
			if ((ballA | ballB) == 0)
				break;
			const uint32_t cands = group_bits(ballA, g) | (group_bits(ballB, g) << 16); // bit b - 1
			const bool tryit = cands != 0;
			const uint32_t bL = uint32_t(__builtin_ctz(cands | 0x80000000u)) + 1u;
			candA = candA && bA != bL;
			candB = candB && bB != bL;
			const uint4 par = pwg[tryit ? bL : 1u];
			uint32_t required, how;
			evaluate(bL, par, required, how);
			if (tryit && required < best) {
				best = required;
				got = how;
				gotpar = par.z;
			}
		}
	}
	if (got) {
		// the last width adopted, :365-382
		const uint32_t bL = got & 0xffu, xm = gotpar >> 24;
		sel.encoding = OROCH_BITPFR;
		sel.origin = mn;
		sel.nbits = bL;
		sel.noutliers = gotpar & 0xffffu;
		sel.venc = (gotpar >> 16) & 0xffu;
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
	}
}

__device__ __forceinline__ uint32_t stream_size(const quad_selection &sel)
{
	return (((sel.metaspace + 7u) & ~7u) + sel.dataspace + 15u) & ~15u;
}

// The descriptor halves in put_desc's layout.
__device__ __forceinline__ void desc_words(const quad_selection &sel, uint4 &da, uint4 &db)
{
	da = make_uint4(sel.origin, 0u, sel.dataspace, stream_size(sel));
	db = make_uint4(sel.metaspace | (sel.noutliers << 16),
			uint32_t(BLOCK) | (sel.encoding << 16) | (sel.nbits << 24),
			sel.ienc | (sel.inbits << 8) | (sel.venc << 16) | (sel.vnbits << 24), 0u);
}

#ifndef SELECT_QUAD_WAVES
#define SELECT_QUAD_WAVES 5
#endif
// (registers capped at what five wavefronts per SIMD allow: the kernel is
// bound by instruction issue and the fifth wavefront fills stalls of the others)
template <typename E>
__global__ __launch_bounds__(WAVE *WAVES_PER_WG) __attribute__((amdgpu_waves_per_eu(SELECT_QUAD_WAVES, 8))) void k_select_quad(const typename E::S *__restrict__ src,
								      size_t n, size_t nblocks,
								      oroch_block_desc *__restrict__ desc,
								      uint32_t *__restrict__ sizes)
{
	using C = typename E::C;
	__shared__ __attribute__((aligned(16))) uint32_t hqs[WAVES_PER_WG][QUAD * HQ_DWORDS];
	__shared__ __attribute__((aligned(16))) uint32_t nos[WAVES_PER_WG][QUAD * NO_DWORDS];
	__shared__ __attribute__((aligned(16))) uint4 pws[WAVES_PER_WG][QUAD * PW_ENTRIES];
	__shared__ uint4 wt[65];
	const int wave = uniform(int(threadIdx.x / WAVE)), lane = lane_id();
	load_width_table(wt);
	const lane_consts lc = make_lane_consts(wt, lane);
	const quad_lane_consts qlc = make_quad_lane_consts(wt, lane);
	const size_t nquads = (nblocks + QUAD - 1) / QUAD;
	const size_t stride = size_t(gridDim.x) * WAVES_PER_WG;
	for (size_t q = size_t(blockIdx.x) * WAVES_PER_WG + wave; q < nquads; q += stride) {
		const size_t blk0 = q * QUAD;
		if (blk0 + QUAD <= nblocks && (blk0 + QUAD) * BLOCK <= n) {
			const int g = int(uint32_t(lane) >> 4), sub = int(uint32_t(lane) & 15u);
			const size_t blk = blk0 + size_t(g);
			C v[VPQ];
			load16<E>(src + blk * BLOCK + size_t(sub) * VPQ, v);
			quad_selection sel;
			select_quad<E>(v, lane, hqs[wave], nos[wave], pws[wave], wt, qlc, sel);
			uint4 da, db;
			desc_words(sel, da, db);
			if (desc && sub < 2) {
				// (component-wise: selecting whole vectors by lane sends them through scratch)
				const uint4 half = make_uint4(sub ? db.x : da.x, sub ? db.y : da.y, sub ? db.z : da.z,
							      sub ? db.w : da.w);
				reinterpret_cast<uint4 *>(desc + blk)[sub] = half;
			}
			if (sizes && sub == 0)
				sizes[blk] = da.w;
		} else {
			for (int b = 0; b < QUAD && blk0 + size_t(b) < nblocks; b++) {
				const size_t blk = blk0 + size_t(b);
				const int nv = block_count(n, blk);
				C v[VPL];
				load_block<E>(src, blk * BLOCK, nv, lane, v);
				selection<C> sel;
				select_any<E>(v, nv, lane, hqs[wave], wt, lc, sel);
				if (lane == 0) {
					if (desc)
						put_desc(&desc[blk], sel, nv);
					if (sizes)
						sizes[blk] = stream_size(sel);
				}
				wave_sync();
			}
		}
		wave_sync();
	}
}

} // namespace oroch_hip
