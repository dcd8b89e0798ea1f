// group.hip -- per-block statistics, codec selection, batch encode and decode.
//
// One 64-lane wavefront owns one block of up to 256 values; lane l holds the
// four consecutive values 4l .. 4l+3 (one 16-byte load per lane for 32-bit
// types).  A 256-thread workgroup is four independent wavefronts.
//
// Reference behaviour reproduced here (paths below /root/reference/oroch):
//   integer_stats.h:48-53, 74-80, 97-102   min / max / bit-length histogram
//   integer_codec.h:234-384, 405-496       select (incl. the estimate gate)
//   integer_codec.h:128-214                metadata bytes
//   integer_codec.h:386-402, 498-609       encode / decode dispatch, bitpfr lists
//   integer_group.h:44-76                  [meta][pad to 8][data] framing
//   bitpck.h:73-201, varint.h:74-115       16-byte block packing, varints
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "bitfield.hpp"
#include "scan.hpp"
#include "block.hpp"

// Most workgroups a launch of each block kernel asks for (grid_for).  Found by
// measurement: a wavefront should have a few tiles (blocks, or quads of blocks)
// to walk, so that the fetches ahead of the walk overlap with work; beyond
// that, more and shorter-lived workgroups let the dispatcher even things out,
// up to a point that differs per kernel -- the decoder of four blocks per
// wavefront, which keeps offsets and stream bytes of two tiles in flight, and
// the selection want longer walks than the encoders.
#ifndef GRID_SELECT_QUAD
#define GRID_SELECT_QUAD 16384u
#endif
#ifndef GRID_ENCODE_QUAD
#define GRID_ENCODE_QUAD 65536u
#endif
#ifndef GRID_DECODE_QUAD
#define GRID_DECODE_QUAD 16384u
#endif
#ifndef GRID_SELECT
#define GRID_SELECT 131072u
#endif
#ifndef GRID_ENCODE
#define GRID_ENCODE 262144u
#endif
#ifndef GRID_DECODE
#define GRID_DECODE 262144u
#endif

namespace oroch_hip {

// Wave-uniform outcome of selection for one block.
template <typename C>
struct selection
{
	uint32_t encoding, nbits;
	uint32_t metaspace;  // including the encoding byte
	uint32_t dataspace;
	uint32_t noutliers;
	uint32_t ienc, inbits, venc, vnbits;
	C origin;
};

// ---- exception positions ------------------------------------------------------
//
// For a width b, the exceptions of a block are the values whose delta needs
// more than b bits.  prev_exception() gives every lane the position of the last
// exception in lower lanes (-1 if none): offset.h:55-60 with offset 1 stores
// position - previous - 1, the first one relative to -1.
__device__ __forceinline__ int prev_exception(const bool (&f)[VPL], int lane)
{
	int lp = -1;
#pragma unroll
	for (int s = 0; s < VPL; s++)
		if (f[s])
			lp = lane * VPL + s;
	int incl = wave_scan_max(lp, lane);
	int ex = __shfl_up(incl, 1, WAVE);
	return lane == 0 ? -1 : ex;
}

// ---- selection ----------------------------------------------------------------

// integer_codec.h:234-384 for the block held in v[] (nv valid values).
// hist: 65 dwords of LDS owned by this wavefront.
// FULL: the block holds all 256 values (every lane's four are valid).
template <typename E, bool FULL>
__device__ void select_block(const typename E::C (&v)[VPL], int nv, int lane, uint32_t *hist,
			     const uint4 *wt, selection<typename E::C> &sel)
{
	using C = typename E::C;

	sel.encoding = OROCH_NORMAL;
	sel.nbits = 0;
	sel.metaspace = 1;
	sel.dataspace = 0;
	sel.noutliers = 0;
	sel.ienc = sel.venc = OROCH_NORMAL;
	sel.inbits = sel.vnbits = 0;
	sel.origin = 0;

	// integer_codec.h:247-253
	if (nv == 0)
		return;

	// integer_stats.h:88-95 on order-preserving unsigned keys
	C kmin = ~C(0), kmax = 0;
#pragma unroll
	for (int s = 0; s < VPL; s++) {
		if (FULL || lane * VPL + s < nv) {
			C k = v[s] ^ E::FLIP;
			kmin = k < kmin ? k : kmin;
			kmax = k > kmax ? k : kmax;
		}
	}
	const C mn = wave_min(kmin) ^ E::FLIP;
	const C mx = wave_max(kmax) ^ E::FLIP;
	const uint32_t vo = vsp(zz<E>(mn)); // varint bytes of the origin

	// integer_codec.h:255-264
	if (mn == mx) {
		sel.encoding = OROCH_NAUGHT;
		sel.origin = mn;
		sel.metaspace = 1 + vo;
		return;
	}

	// deltas, their bit lengths, and the two varint footprints
	// (integer_stats.h:97-102, integer_codec.h:483-489)
	uint32_t ud[VPL];
	uint32_t sums = 0;
#pragma unroll
	for (int s = 0; s < VPL; s++) {
		const bool ok = FULL || lane * VPL + s < nv;
		C d = (v[s] - mn) & E::MASK;
		ud[s] = bitlen(d);
		if (ok)
			sums += vsp(zz<E>(v[s])) + (vsp_bits(ud[s]) << 16);
	}
	sums = wave_sum(sums);
	const uint32_t vis = sums & 0xffffu, vfs = sums >> 16;

	// select_basic, integer_codec.h:422-496; strictly smaller replaces (:413)
	uint32_t best = uint32_t(nv) * E::BYTES; // data + meta (without encoding byte)
	uint32_t bmeta0 = 0;
	{
		C zmax = mx;
		if constexpr (E::sgn) {
			C a = zz<E>(mn), b = zz<E>(mx);
			zmax = a > b ? a : b;
		}
		uint32_t nb = bitlen(zmax);
		uint32_t total = pck_space(wt, nv, nb) + 1;
		if (total < best) {
			best = total;
			bmeta0 = 1;
			sel.encoding = OROCH_BITPCK;
			sel.nbits = nb;
			sel.origin = 0;
		}
	}
	const uint32_t nbmax = bitlen(C((mx - mn) & E::MASK));
	{
		uint32_t total = pck_space(wt, nv, nbmax) + 1 + vo;
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

	// integer_codec.h:271-272
	if (nv < 5)
		return;

	// ---- patched frame of reference, integer_codec.h:280-383 ----
	// histogram of bit lengths into LDS, then lane l looks after width b = l + 1
	hist[lane] = 0;
	if (lane == 0)
		hist[64] = 0;
	wave_sync();
#pragma unroll
	for (int s = 0; s < VPL; s++)
		if (FULL || lane * VPL + s < nv)
			atomicAdd(&hist[ud[s]], 1u);
	wave_sync();
	const uint32_t b = uint32_t(lane) + 1;
	const uint32_t hb = hist[b];
	wave_sync();

	// values longer than b bits: all non-zero deltas minus those of 1..b bits
	const uint32_t upto = wave_scan_sum(hb, lane);
	const uint32_t nout = read_lane(upto, WAVE - 1) - upto;
	// varint bytes of the high parts: sum over j >= 0 of nout(b + 7j)
	// (each value of bit length k > b costs ceil((k - b) / 7) bytes, :308-312)
	uint32_t valvar = nout;
	valvar += shfl_down_zero(valvar, 7, lane);
	valvar += shfl_down_zero(valvar, 14, lane);
	valvar += shfl_down_zero(valvar, 28, lane);
	valvar += shfl_down_zero(valvar, 56, lane);

	const bool active = b < nbmax && hb != 0;
	const uint32_t basic_meta = 1 + vo;
	const uint32_t bw = active ? b : 1u;
	const uint32_t xmeta = 2 + vsp_bits(bitlen(nout));
	const uint32_t valpck = pck_space(wt, nout, active ? nbmax - b : 1u);
	const uint32_t venc = valpck < valvar ? OROCH_BITPCK : OROCH_VARINT;
	const uint32_t vdata = valpck < valvar ? valpck : valvar;
	uint32_t indmin = pck_space(wt, nout, 1);
	indmin = indmin > uint32_t(nv) ? uint32_t(nv) : indmin;
	const uint32_t fixed = basic_meta + xmeta + pck_space(wt, nv, bw) + vdata;
	const uint32_t est = fixed + indmin;

	// Replay of the reference's ascending loop.  `best` only shrinks, so a
	// width failing the gate (:327-335) against the initial value fails later.
	uint64_t cand = __ballot(active && est < best);
	while (cand) {
		const int L = __builtin_ctzll(cand);
		cand &= cand - 1;
		const uint32_t estL = uint32_t(__builtin_amdgcn_readlane(int(est), L));
		if (estL >= best)
			continue;
		const uint32_t bL = uint32_t(L) + 1;
		const uint32_t noutL = uint32_t(__builtin_amdgcn_readlane(int(nout), L));

		// index gaps of this width's exceptions, :338-351
		bool f[VPL];
#pragma unroll
		for (int s = 0; s < VPL; s++)
			f[s] = (FULL || lane * VPL + s < nv) && ud[s] > bL;
		int prev = prev_exception(f, lane);
		uint32_t gmax = 0, gsum = 0;
#pragma unroll
		for (int s = 0; s < VPL; s++) {
			if (f[s]) {
				int pos = lane * VPL + s;
				uint32_t gap = uint32_t(pos - prev - 1);
				prev = pos;
				gmax = gap > gmax ? gap : gmax;
				gsum += vsp_bits(bitlen(gap));
			}
		}
		// one reduction for both: max in the high half, sum in the low half
		// does not combine, so two passes
		gmax = wave_max(gmax);
		const uint32_t indvar = wave_sum(gsum);
		uint32_t indnbits = bitlen(gmax);
		indnbits = indnbits ? indnbits : 1u;
		const uint32_t indpck = pck_space(wt, noutL, indnbits);
		const uint32_t ienc = indpck < indvar ? OROCH_BITPCK : OROCH_VARINT;
		const uint32_t idata = indpck < indvar ? indpck : indvar;

		const uint32_t fixedL = uint32_t(__builtin_amdgcn_readlane(int(fixed), L));
		const uint32_t required = fixedL + idata;
		if (required < best) {
			best = required;
			const uint32_t xm = uint32_t(__builtin_amdgcn_readlane(int(xmeta), L));
			sel.encoding = OROCH_BITPFR;
			sel.origin = mn;
			sel.nbits = bL;
			sel.noutliers = noutL;
			sel.venc = uint32_t(__builtin_amdgcn_readlane(int(venc), L));
			sel.vnbits = nbmax - bL;
			sel.ienc = ienc;
			sel.inbits = ienc == OROCH_BITPCK ? indnbits : 0u;
			sel.metaspace = 1 + basic_meta + xm;
			sel.dataspace = required - basic_meta - xm;
		}
	}
}

// ---- selection, full blocks -----------------------------------------------------
//
// Same rule as select_block for a block of exactly 256 values, arranged to
// keep the vector instruction count down:
//  * both bit-length histograms (of v - min and of zigzag(v)) are built with LDS
//    atomics; the two varint footprints (integer_codec.h:485-489) and the two
//    bit-packing widths (:441-452, :465-468) then come from the histograms, one
//    bin per lane, instead of from per-value arithmetic and a max reduction;
//  * per-lane constants (bytes of a varint of lane + 1 bits, space of 256 values
//    at lane + 1 bits) are computed once per kernel;
//  * the index-gap walk (:338-351) is skipped where its outcome is forced:
//    with at most one value inside the width every gap is 0 or 1, and with at
//    most 14 exceptions the varint list (<= 15 bytes) always beats a packed one
//    (>= 16 bytes), so only "is some gap >= 128" is needed.
//    In that last case index_nbits is left 0: it is not part of the stream
//    when the index list is a varint list (integer_codec.h:177-182).

struct lane_consts
{
	uint32_t w;       // varint bytes of a value of bit length lane + 1
	uint32_t pck256;  // bitpck space of 256 values at lane + 1 bits
};

__device__ __forceinline__ lane_consts make_lane_consts(const uint4 *wt, int lane)
{
	lane_consts lc;
	lc.w = vsp_bits(uint32_t(lane) + 1);
	lc.pck256 = wt[lane + 1].w;
	return lc;
}

// hq: 65 x 4 dwords of LDS owned by this wavefront (16-byte aligned).
// wt: {capacity, magic} per width, shared by the workgroup (width_table in LDS).
// ud[]: bit lengths of the deltas, for the caller.
template <typename E>
__device__ void select_full(const typename E::C (&v)[VPL], int lane, uint32_t *hq, const uint4 *wt,
			    const lane_consts &lc, selection<typename E::C> &sel)
{
	using C = typename E::C;

	sel.encoding = OROCH_NORMAL;
	sel.nbits = 0;
	sel.metaspace = 1;
	sel.dataspace = 0;
	sel.noutliers = 0;
	sel.ienc = sel.venc = OROCH_NORMAL;
	sel.inbits = sel.vnbits = 0;
	sel.origin = 0;

	C kmin = v[0] ^ E::FLIP;
#pragma unroll
	for (int s = 1; s < VPL; s++) {
		const C k = v[s] ^ E::FLIP;
		kmin = k < kmin ? k : kmin;
	}
	const C mn = wave_min(kmin) ^ E::FLIP;
	const uint32_t vo = vsp(zz<E>(mn));

	// Bin k of the table is four dwords, one per lane & 3: the copies cut
	// same-address serialisation of the LDS atomics by four.  The zigzag images
	// need no histogram: selection wants the sum of their varint lengths
	// (ceil(L / 7) = (37 L + 219) >> 8 for a bit length L of 1..64; zero takes
	// one byte like L = 1) and the longest of them.
	uint32_t ud[VPL];
	uint32_t zbytes = 0, zlong = 1;
	*reinterpret_cast<uint4 *>(&hq[lane * 4]) = make_uint4(0, 0, 0, 0);
	if (lane == 0)
		*reinterpret_cast<uint4 *>(&hq[256]) = make_uint4(0, 0, 0, 0);
	wave_sync();
	uint32_t *mine = hq + (lane & 3);
#pragma unroll
	for (int s = 0; s < VPL; s++) {
		ud[s] = bitlen(C((v[s] - mn) & E::MASK));
		atomicAdd(&mine[ud[s] * 4], 1u);
		uint32_t zl = bitlen(zz<E>(v[s]));
		zl = zl ? zl : 1u;
		zbytes += (37u * zl + 219u) >> 8;
		zlong = zl > zlong ? zl : zlong;
	}
	wave_sync();
	const uint4 q = *reinterpret_cast<const uint4 *>(&hq[(lane + 1) * 4]);
	const uint4 q0 = *reinterpret_cast<const uint4 *>(&hq[0]);
	wave_sync();
	const uint32_t qs = q.x + q.y + q.z + q.w, q0s = uniform(q0.x + q0.y + q0.z + q0.w);
	const uint2 h = make_uint2(qs, 0);
	const uint32_t zero_d = q0s;

	// integer_codec.h:255-264
	if (zero_d == BLOCK) {
		sel.encoding = OROCH_NAUGHT;
		sel.origin = mn;
		sel.metaspace = 1 + vo;
		return;
	}

	const uint32_t t = wave_sum(h.x * lc.w + (zbytes << 16));
	const uint32_t vfs = (t & 0xffffu) + zero_d, vis = t >> 16;
	const uint32_t nbmax = 64u - uint32_t(__builtin_clzll(__ballot(h.x != 0)));
	const uint32_t nbz = wave_max(zlong);

	// select_basic, integer_codec.h:422-496
	uint32_t best = BLOCK * E::BYTES, bmeta0 = 0;
	{
		const uint32_t total = read_lane(lc.pck256, int(nbz) - 1) + 1;
		if (total < best) {
			best = total;
			bmeta0 = 1;
			sel.encoding = OROCH_BITPCK;
			sel.nbits = nbz;
		}
	}
	{
		const uint32_t total = read_lane(lc.pck256, int(nbmax) - 1) + 1 + vo;
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

	// ---- patched frame of reference, integer_codec.h:280-383; lane l <-> width l + 1
	const uint32_t b = uint32_t(lane) + 1;
	const uint32_t upto = wave_scan_sum(h.x, lane);
	const uint32_t nout = read_lane(upto, WAVE - 1) - upto;
	uint32_t valvar = nout;
	valvar += shfl_down_zero(valvar, 7, lane);
	valvar += shfl_down_zero(valvar, 14, lane);
	valvar += shfl_down_zero(valvar, 28, lane);
	valvar += shfl_down_zero(valvar, 56, lane);

	const bool active = b < nbmax && h.x != 0;
	const uint32_t basic_meta = 1 + vo;
	const uint32_t xmeta = 2 + (nout < 128u ? 1u : 2u);
	const uint4 cm = wt[active ? nbmax - b : 1u];
	const uint32_t valpck = 16u * __umulhi(nout + cm.x - 1u, cm.y);
	const uint32_t venc = valpck < valvar ? OROCH_BITPCK : OROCH_VARINT;
	const uint32_t vdata = valpck < valvar ? valpck : valvar;
	const uint32_t indmin = nout > 128u ? 32u : 16u; // min(pck(nout, 1), 256), nout in 1..255
	const uint32_t fixed = basic_meta + xmeta + lc.pck256 + vdata;
	const uint32_t est = fixed + indmin;

	// What width bL = L + 1 needs once its index list is known (:338-351).
	auto evaluate = [&](int L, uint32_t &required, uint32_t &ienc, uint32_t &indnbits) {
		const uint32_t bL = uint32_t(L) + 1;
		const uint32_t noutL = read_lane(nout, L);
		uint32_t indvar;
		if (noutL >= BLOCK - 1) {
			indnbits = 1;
			indvar = noutL;
		} else {
			// last exception at or before each lane, then the gaps
			bool f[VPL];
#pragma unroll
			for (int s = 0; s < VPL; s++)
				f[s] = ud[s] > bL;
			int prev = prev_exception(f, lane);
			if (noutL <= 14) {
				// only the gap in front of each lane's first exception can be long
				int first = f[0] ? 0 : f[1] ? 1 : f[2] ? 2 : 3;
				const bool any = f[0] | f[1] | f[2] | f[3];
				const bool big = any && lane * VPL + first - prev - 1 >= 128;
				indnbits = 0;
				indvar = noutL + (__ballot(big) ? 1u : 0u);
			} else {
				uint32_t gmax = 0;
#pragma unroll
				for (int s = 0; s < VPL; s++) {
					const int pos = lane * VPL + s;
					const uint32_t gap = uint32_t(pos - prev - 1);
					gmax = f[s] && gap > gmax ? gap : gmax;
					prev = f[s] ? pos : prev;
				}
				gmax = wave_max(gmax);
				indnbits = bitlen(gmax);
				indnbits = indnbits ? indnbits : 1u;
				// 256 positions hold at most one gap of 128 or more
				indvar = noutL + (gmax >= 128u ? 1u : 0u);
			}
		}
		uint32_t indpck = ~0u;
		if (indnbits) {
			const uint4 ci = wt[indnbits];
			indpck = 16u * __umulhi(noutL + uniform(ci.x) - 1u, uniform(ci.y));
		}
		ienc = indpck < indvar ? OROCH_BITPCK : OROCH_VARINT;
		const uint32_t idata = indpck < indvar ? indpck : indvar;
		required = read_lane(fixed, L) + idata;
	};
	auto adopt = [&](int L, uint32_t required, uint32_t ienc, uint32_t indnbits) {
		best = required;
		const uint32_t xm = read_lane(xmeta, L);
		sel.encoding = OROCH_BITPFR;
		sel.origin = mn;
		sel.nbits = uint32_t(L) + 1;
		sel.noutliers = read_lane(nout, L);
		sel.venc = read_lane(venc, L);
		sel.vnbits = nbmax - uint32_t(L) - 1;
		sel.ienc = ienc;
		sel.inbits = ienc == OROCH_BITPCK ? indnbits : 0u;
		sel.metaspace = 1 + basic_meta + xm;
		sel.dataspace = required - basic_meta - xm;
	};

	// The reference walks the widths upwards, looks at one when its estimate is
	// below the best total so far (:327-335) and adopts it when its real need is
	// below that too.  One evaluation replaces the walk when the width with the
	// lowest lower bound (its index list at its shortest: one byte per exception
	// or the one-bit packing) comes out, estimate and real need, below every
	// other candidate's lower bound and below the best basic encoding: nothing
	// before it can then have hidden it, nothing after it can beat it.
	const bool c0 = active && est < best;
	uint64_t cand = __ballot(c0);
	if (cand == 0)
		return;
	const uint32_t low = fixed + (nout < indmin ? nout : indmin);
	const uint32_t key = c0 ? (low << 6) | uint32_t(lane) : ~0u;
	const uint32_t first = wave_min(key);
	const uint32_t second = wave_min(key == first ? ~0u : key);
	const int L1 = int(first & 63u);
	uint32_t required1, ienc1, indnbits1;
	evaluate(L1, required1, ienc1, indnbits1);
	{
		const uint32_t estL = read_lane(est, L1);
		const uint32_t top = required1 > estL ? required1 : estL;
		const uint32_t bound = best < (second >> 6) ? best : (second >> 6);
		if (top < bound) {
			adopt(L1, required1, ienc1, indnbits1);
			return;
		}
	}
	// the walk itself; what is known of width L1 + 1 is used again
	while (cand) {
		const int L = __builtin_ctzll(cand);
		cand &= cand - 1;
		if (read_lane(est, L) >= best)
			continue;
		uint32_t required = required1, ienc = ienc1, indnbits = indnbits1;
		if (L != L1)
			evaluate(L, required, ienc, indnbits);
		if (required < best)
			adopt(L, required, ienc, indnbits);
	}
}

template <typename C>
__device__ __forceinline__ uint32_t stream_size(const selection<C> &sel)
{
	return (((sel.metaspace + 7u) & ~7u) + sel.dataspace + 15u) & ~15u;
}

template <typename C>
__device__ __forceinline__ void put_desc(oroch_block_desc *d, const selection<C> &sel, int nv)
{
	// 32 bytes as two 16-byte stores
	uint32_t w[8];
	w[0] = uint32_t(uint64_t(sel.origin));
	w[1] = uint32_t(uint64_t(sel.origin) >> 32);
	w[2] = sel.dataspace;
	w[3] = stream_size(sel);
	w[4] = sel.metaspace | (sel.noutliers << 16);
	w[5] = uint32_t(nv) | (sel.encoding << 16) | (sel.nbits << 24);
	w[6] = sel.ienc | (sel.inbits << 8) | (sel.venc << 16) | (sel.vnbits << 24);
	w[7] = 0;
	uint4 *q = reinterpret_cast<uint4 *>(d);
	q[0] = make_uint4(w[0], w[1], w[2], w[3]);
	q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// The selection a descriptor holds, from its two 16-byte halves (every lane
// has loaded the same words).
template <typename C>
__device__ __forceinline__ void desc_selection(const uint4 &a, const uint4 &b, selection<C> &sel);

template <typename C>
__device__ __forceinline__ void get_desc(const oroch_block_desc *d, selection<C> &sel)
{
	const uint4 *q = reinterpret_cast<const uint4 *>(d);
	desc_selection(q[0], q[1], sel);
}

template <typename C>
__device__ __forceinline__ void desc_selection(const uint4 &a, const uint4 &b, selection<C> &sel)
{
	sel.origin = C((uint64_t(uniform(a.y)) << 32) | uniform(a.x));
	sel.dataspace = uniform(a.z);
	uint32_t w4 = uniform(b.x), w5 = uniform(b.y), w6 = uniform(b.z);
	sel.metaspace = w4 & 0xffffu;
	sel.noutliers = w4 >> 16;
	sel.encoding = (w5 >> 16) & 0xffu;
	sel.nbits = w5 >> 24;
	sel.ienc = w6 & 0xffu;
	sel.inbits = (w6 >> 8) & 0xffu;
	sel.venc = (w6 >> 16) & 0xffu;
	sel.vnbits = w6 >> 24;
}

// ---- building the block image in LDS --------------------------------------------

// varint.h:74-83 into a byte image; returns the new position.
__device__ __forceinline__ uint32_t put_varint(uint8_t *img, uint32_t pos, uint64_t u)
{
	while (u >= 0x80) {
		img[pos++] = uint8_t(u | 0x80);
		u >>= 7;
	}
	img[pos++] = uint8_t(u);
	return pos;
}

// Builds [meta][pad to 8][data] (integer_group.h:44-60) for the block in v[]
// into `img` (LDS, 16-byte aligned, zero filled here) and returns its length.
template <typename E, bool FULL>
__device__ uint32_t encode_block(const typename E::C (&v)[VPL], int nv, int lane,
				 const selection<typename E::C> &sel, const uint4 *wt, uint32_t *img)
{
	using C = typename E::C;
	uint8_t *img8 = reinterpret_cast<uint8_t *>(img);
	const uint32_t doff = (sel.metaspace + 7u) & ~7u;
	const uint32_t total = doff + sel.dataspace;
	const uint32_t total16 = (total + 15u) & ~15u;

	for (uint32_t o = uint32_t(lane) * 16u; o < total16; o += WAVE * 16u)
		*reinterpret_cast<uint4 *>(img8 + o) = make_uint4(0, 0, 0, 0);
	wave_sync();

	// metadata, integer_codec.h:128-151, 177-182, 196-204: at most 16 bytes,
	// assembled in two scalar registers and stored by one lane
	{
		uint64_t h0 = sel.encoding, h1 = 0;
		uint32_t p = 1;
		auto put = [&](uint32_t byte) {
			const uint64_t x = uint64_t(byte) << ((p & 7u) * 8u);
			if (p & 8u)
				h1 |= x;
			else
				h0 |= x;
			p++;
		};
		auto put_var = [&](uint64_t u) {
			while (u >= 0x80) {
				put(uint32_t(u & 0x7f) | 0x80u);
				u >>= 7;
			}
			put(uint32_t(u));
		};
		switch (sel.encoding) {
		case OROCH_NAUGHT:
		case OROCH_VARFOR:
			put_var(uint64_t(zz<E>(sel.origin)));
			break;
		case OROCH_BITPFR:
		case OROCH_BITFOR:
			put_var(uint64_t(zz<E>(sel.origin)));
			put(sel.nbits);
			break;
		case OROCH_BITPCK:
			put(sel.nbits);
			break;
		default:
			break;
		}
		if (sel.encoding == OROCH_BITPFR) {
			put_var(sel.noutliers);
			put(sel.ienc == OROCH_BITPCK ? sel.inbits : 0u);
			put(sel.venc == OROCH_BITPCK ? sel.vnbits : 0u);
		}
		if (lane == 0) {
			uint64_t *img64 = reinterpret_cast<uint64_t *>(img);
			img64[0] = h0;
			if (p > 8)
				img64[1] = h1;
		}
	}

	const uint32_t enc = sel.encoding;
	if (enc == OROCH_NORMAL) {
		// normal.h:42-49
		typename E::S *raw = reinterpret_cast<typename E::S *>(img8 + doff);
#pragma unroll
		for (int s = 0; s < VPL; s++)
			if (FULL || lane * VPL + s < nv)
				raw[lane * VPL + s] = typename E::S(v[s]);
	} else if (enc == OROCH_BITPCK || enc == OROCH_BITFOR || enc == OROCH_BITPFR) {
		// bitpck.h:73-128: value k of a 16-byte block at bit k * nbits
		const uint32_t nb = sel.nbits;
		const uint32_t cap = uniform(wt[nb].x), magic = uniform(wt[nb].y);
		const C low = nb >= uint32_t(E::W) ? E::MASK : C((C(1) << (nb % E::W)) - 1);
		C u[VPL];
		bool f[VPL];
		// (a field of a 64-bit element that is no wider than 32 bits goes in as a
		// 32-bit one: two words instead of three, 32-bit shifts)
		const bool narrow = sizeof(C) == 8 && nb <= 32u;
#pragma unroll
		for (int s = 0; s < VPL; s++) {
			const bool ok = FULL || lane * VPL + s < nv;
			u[s] = enc == OROCH_BITPCK ? zz<E>(v[s]) : C((v[s] - sel.origin) & E::MASK);
			f[s] = ok && enc == OROCH_BITPFR && (u[s] & ~low) != 0;
			if (ok) {
				const uint32_t pos = uint32_t(lane * VPL + s);
				const uint32_t blk = __umulhi(pos, magic);
				const uint32_t k = pos - blk * cap;
				const uint32_t bit = doff * 8u + blk * 128u + k * nb;
				if (narrow)
					lds_or_bits(img, bit, uint32_t(u[s] & low));
				else
					lds_or_bits(img, bit, C(u[s] & low));
			}
		}
		if (enc == OROCH_BITPFR) {
			// exception lists, integer_codec.h:573-580: indices then values,
			// each bit-packed (element types size_t / unsigned T) or varint
			int prev = prev_exception(f, lane);
			uint32_t gap[VPL];
			C hv[VPL];
			uint32_t acc = 0; // rank | index bytes << 10 | value bytes << 20
#pragma unroll
			for (int s = 0; s < VPL; s++) {
				gap[s] = 0;
				hv[s] = 0;
				if (f[s]) {
					int pos = lane * VPL + s;
					gap[s] = uint32_t(pos - prev - 1);
					prev = pos;
					hv[s] = u[s] >> nb;
					acc += 1u + (vsp_bits(bitlen(gap[s])) << 10) + (vsp(hv[s]) << 20);
				}
			}
			uint32_t run = wave_scan_sum(acc, lane) - acc; // exclusive
			const uint32_t ibase = doff + pck_space(wt, uint32_t(nv), nb);
			const uint32_t ilen = sel.ienc == OROCH_BITPCK
						      ? pck_space(wt, sel.noutliers, sel.inbits)
						      : (wave_sum(acc) >> 10) & 0x3ffu;
			const uint32_t vbase = ibase + ilen;
			const uint4 iw = wt[sel.inbits ? sel.inbits : 1u], vw = wt[sel.vnbits ? sel.vnbits : 1u];
			const uint32_t icap = uniform(iw.x), imagic = uniform(iw.y);
			const uint32_t vcap = uniform(vw.x), vmagic = uniform(vw.y);
#pragma unroll
			for (int s = 0; s < VPL; s++) {
				if (f[s]) {
					const uint32_t e = run & 0x3ffu;
					if (sel.ienc == OROCH_BITPCK) {
						const uint32_t blk = __umulhi(e, imagic);
						lds_or_bits(img, ibase * 8u + blk * 128u + (e - blk * icap) * sel.inbits,
							    gap[s]);
					} else {
						put_varint(img8, ibase + ((run >> 10) & 0x3ffu), gap[s]);
					}
					if (sel.venc == OROCH_BITPCK) {
						const uint32_t blk = __umulhi(e, vmagic);
						lds_or_bits(img, vbase * 8u + blk * 128u + (e - blk * vcap) * sel.vnbits,
							    hv[s]);
					} else {
						put_varint(img8, vbase + (run >> 20), uint64_t(hv[s]));
					}
					run += 1u + (vsp_bits(bitlen(gap[s])) << 10) + (vsp(hv[s]) << 20);
				}
			}
		}
	} else if (enc == OROCH_VARINT || enc == OROCH_VARFOR) {
		// varint.h:133-139 with zigzag (varint) or origin_codec (varfor)
		C u[VPL];
		uint32_t len = 0;
#pragma unroll
		for (int s = 0; s < VPL; s++) {
			u[s] = enc == OROCH_VARINT ? zz<E>(v[s]) : C((v[s] - sel.origin) & E::MASK);
			if (FULL || lane * VPL + s < nv)
				len += vsp(u[s]);
		}
		uint32_t at = doff + wave_scan_sum(len, lane) - len;
#pragma unroll
		for (int s = 0; s < VPL; s++)
			if (FULL || lane * VPL + s < nv)
				at = put_varint(img8, at, uint64_t(u[s]));
	}
	wave_sync();
	return total;
}

// Copies an LDS image of `len` bytes to 16-byte aligned global memory.
__device__ __forceinline__ void flush_image(const uint32_t *img, uint32_t len, uint8_t *dst, int lane)
{
	const uint32_t len16 = (len + 15u) & ~15u;
	const uint8_t *img8 = reinterpret_cast<const uint8_t *>(img);
	for (uint32_t o = uint32_t(lane) * 16u; o < len16; o += WAVE * 16u)
		*reinterpret_cast<uint4 *>(dst + o) = *reinterpret_cast<const uint4 *>(img8 + o);
}

// ---- kernels ----------------------------------------------------------------------

// Selection for the block in v[]: the lean path for full blocks, the general
// one for a short last block.
template <typename E>
__device__ __forceinline__ void select_any(const typename E::C (&v)[VPL], int nv, int lane,
					   uint32_t *hq, const uint4 *wt, const lane_consts &lc,
					   selection<typename E::C> &sel)
{
	if (nv == BLOCK)
		select_full<E>(v, lane, hq, wt, lc, sel);
	else
		select_block<E, false>(v, nv, lane, hq, wt, sel);
}

// One block per wavefront.  The kernel is bound by the latency of its loads (a
// constant block takes as long as any other), so the values of a wavefront's
// next block are asked for before the current one is looked at; the loads are
// unconditional (the last full block is asked for again where there is no next
// one) and the stores of a block are the same three instructions on every path,
// so that waiting for the values does not mean waiting for the stores behind
// them.  A short last block is taken after the loop.
template <typename E>
__device__ __forceinline__ void load_full(const typename E::S *__restrict__ src, size_t blk, int lane,
					  typename E::C (&v)[VPL])
{
	using C = typename E::C;
	const typename E::S *p = src + blk * BLOCK + size_t(lane) * VPL;
	if constexpr (E::BYTES == 8) {
		const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(p);
		const ulonglong2 a = q[0], b = q[1];
		v[0] = a.x, v[1] = a.y, v[2] = b.x, v[3] = b.y;
	} else if constexpr (E::BYTES == 4) {
		const uint4 a = *reinterpret_cast<const uint4 *>(p);
		v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w;
	} else if constexpr (E::BYTES == 2) {
		const uint2 a = *reinterpret_cast<const uint2 *>(p);
		v[0] = C(a.x & 0xffffu), v[1] = C(a.x >> 16), v[2] = C(a.y & 0xffffu), v[3] = C(a.y >> 16);
	} else {
		const uint32_t a = *reinterpret_cast<const uint32_t *>(p);
		v[0] = C(a & 0xffu), v[1] = C((a >> 8) & 0xffu), v[2] = C((a >> 16) & 0xffu), v[3] = C(a >> 24);
	}
}

template <typename E, bool DESC, bool SIZES>
__global__ __launch_bounds__(WAVE *WAVES_PER_WG) void k_select(const typename E::S *__restrict__ src, size_t n,
								 size_t nblocks, oroch_block_desc *__restrict__ desc,
								 uint32_t *__restrict__ sizes)
{
	using C = typename E::C;
	__shared__ __attribute__((aligned(16))) uint32_t hist2[WAVES_PER_WG][264];
	__shared__ uint4 wt[65];
	const int wave = uniform(int(threadIdx.x / WAVE)), lane = lane_id();
	load_width_table(wt);
	const lane_consts lc = make_lane_consts(wt, lane);
	const size_t stride = size_t(gridDim.x) * WAVES_PER_WG;
	const size_t nfull = n / BLOCK;
	size_t blk = size_t(blockIdx.x) * WAVES_PER_WG + wave;
	auto put = [&](size_t at, const selection<C> &sel, int nv) {
		// Every lane stores the same words to the same place: a store under a
		// lane condition is branched around, and with a branch the wait for the
		// next block's values would be a wait for everything, these stores
		// included.
		const uint4 a = make_uint4(uint32_t(uint64_t(sel.origin)), uint32_t(uint64_t(sel.origin) >> 32),
					   sel.dataspace, stream_size(sel));
		const uint4 b = make_uint4(sel.metaspace | (sel.noutliers << 16),
					   uint32_t(nv) | (sel.encoding << 16) | (sel.nbits << 24),
					   sel.ienc | (sel.inbits << 8) | (sel.venc << 16) | (sel.vnbits << 24), 0u);
		if constexpr (DESC) {
			uint4 *q = reinterpret_cast<uint4 *>(&desc[at]);
			q[0] = a;
			q[1] = b;
		}
		if constexpr (SIZES)
			sizes[at] = a.w;
	};
	if (blk < nfull) {
		// The first block outside the loop, in the loop's own order (values of
		// the next block asked for, this block selected, its descriptor stored):
		// the loop is then entered in the state in which it is repeated, and the
		// wait at its top is for the values alone.
		C v[VPL], vn[VPL];
		load_full<E>(src, blk, lane, v);
		load_full<E>(src, blk + stride < nfull ? blk + stride : blk, lane, vn);
		{
			selection<C> sel;
			select_full<E>(v, lane, hist2[wave], wt, lc, sel);
			put(blk, sel, BLOCK);
		}
		for (blk += stride; blk < nfull; blk += stride) {
#pragma unroll
			for (int s = 0; s < VPL; s++)
				v[s] = vn[s];
			load_full<E>(src, blk + stride < nfull ? blk + stride : blk, lane, vn);
			selection<C> sel;
			select_full<E>(v, lane, hist2[wave], wt, lc, sel);
			put(blk, sel, BLOCK);
		}
	}
	if (blk == nfull && blk < nblocks) {
		const int nv = block_count(n, blk);
		C v[VPL];
		load_block<E>(src, blk * BLOCK, nv, lane, v);
		selection<C> sel;
		select_block<E, false>(v, nv, lane, hist2[wave], wt, sel);
		put(blk, sel, nv);
	}
}

// The image of a full block to the stream in a fixed number of stores per lane,
// whatever its length: a lane whose 16 bytes lie past the image stores the
// image's last 16 bytes again (the same bytes to the same place).  A count of
// stores that does not depend on the block lets the wait for the next block's
// values leave these stores in flight.
template <typename E>
__device__ __forceinline__ void flush_image_fixed(const uint32_t *img, uint32_t len, uint8_t *dst, int lane)
{
	constexpr int ROUNDS = (image_bytes<E>() + WAVE * 16 - 1) / (WAVE * 16);
	const uint32_t last = ((len + 15u) & ~15u) - 16u; // len >= 1
	const uint8_t *img8 = reinterpret_cast<const uint8_t *>(img);
#pragma unroll
	for (int r = 0; r < ROUNDS; r++) {
		uint32_t o = uint32_t(lane) * 16u + uint32_t(r) * (WAVE * 16u);
		o = o < last ? o : last;
		*reinterpret_cast<uint4 *>(dst + o) = *reinterpret_cast<const uint4 *>(img8 + o);
	}
}

// One block per wavefront, as k_select: values, descriptor and stream offset
// of a wavefront's next block are asked for before the current block is
// encoded, by loads that are issued on every path.
template <typename E>
__global__ __launch_bounds__(WAVE *WAVES_PER_WG) void k_encode(const typename E::S *__restrict__ src, size_t n,
								 size_t nblocks,
								 const oroch_block_desc *__restrict__ desc,
								 const uint64_t *__restrict__ offsets,
								 uint8_t *__restrict__ dst)
{
	using C = typename E::C;
	__shared__ __attribute__((aligned(16))) uint32_t image[WAVES_PER_WG][image_bytes<E>() / 4];
	__shared__ uint4 wt[65];
	const int wave = uniform(int(threadIdx.x / WAVE)), lane = lane_id();
	load_width_table(wt);
	const size_t stride = size_t(gridDim.x) * WAVES_PER_WG;
	const size_t nfull = n / BLOCK;
	size_t blk = size_t(blockIdx.x) * WAVES_PER_WG + wave;
	// what is fetched ahead for a block
	struct ahead
	{
		C v[VPL];
		uint4 da, db;
		uint64_t off;
	};
	auto fetch = [&](size_t at, ahead &f) {
		load_full<E>(src, at, lane, f.v);
		const uint4 *q = reinterpret_cast<const uint4 *>(desc + at);
		f.da = q[0];
		f.db = q[1];
		f.off = offsets[at];
	};
	auto code = [&](const ahead &f) {
		selection<C> sel;
		desc_selection(f.da, f.db, sel);
		const uint32_t len = encode_block<E, true>(f.v, BLOCK, lane, sel, wt, image[wave]);
		flush_image_fixed<E>(image[wave], len, dst + uniform(f.off), lane);
		wave_sync();
	};
	if (blk < nfull) {
		// (the first block in the loop's own order, so that the loop is entered in
		// the state in which it is repeated)
		ahead cur, nxt;
		fetch(blk, cur);
		fetch(blk + stride < nfull ? blk + stride : blk, nxt);
		code(cur);
		for (blk += stride; blk < nfull; blk += stride) {
			cur = nxt;
			fetch(blk + stride < nfull ? blk + stride : blk, nxt);
			code(cur);
		}
	}
	if (blk == nfull && blk < nblocks) {
		const int nv = block_count(n, blk);
		C v[VPL];
		load_block<E>(src, blk * BLOCK, nv, lane, v);
		selection<C> sel;
		get_desc(&desc[blk], sel);
		const uint64_t off = uniform(offsets[blk]);
		const uint32_t len = encode_block<E, false>(v, nv, lane, sel, wt, image[wave]);
		flush_image(image[wave], len, dst + off, lane);
	}
}

// ---- block statistics --------------------------------------------------------------
//
// integer_stats.h:48-53 (count, min, max) and :74-80, 97-102 (histogram of the
// bit length of v - min, W + 1 bins) per block of 256 values, one wavefront per
// block: min / max by wavefront reductions, the histogram by LDS atomics.
template <typename E>
__global__ __launch_bounds__(WAVE *WAVES_PER_WG) void k_stats(const typename E::S *src, size_t n,
								size_t nblocks, oroch_block_stats *out)
{
	using C = typename E::C;
	__shared__ uint32_t hist[WAVES_PER_WG][68];
	const int wave = uniform(int(threadIdx.x / WAVE)), lane = lane_id();
	for (size_t blk = size_t(blockIdx.x) * WAVES_PER_WG + wave; blk < nblocks;
	     blk += size_t(gridDim.x) * WAVES_PER_WG) {
		const int nv = block_count(n, blk);
		C v[VPL];
		load_block<E>(src, blk * BLOCK, nv, lane, v);
		C kmin = ~C(0), kmax = 0;
#pragma unroll
		for (int s = 0; s < VPL; s++) {
			if (lane * VPL + s < nv) {
				const C k = v[s] ^ E::FLIP;
				kmin = k < kmin ? k : kmin;
				kmax = k > kmax ? k : kmax;
			}
		}
		const C mn = wave_min(kmin) ^ E::FLIP, mx = wave_max(kmax) ^ E::FLIP;
		hist[wave][lane] = 0;
		if (lane < 4)
			hist[wave][64 + lane] = 0;
		wave_sync();
#pragma unroll
		for (int s = 0; s < VPL; s++)
			if (lane * VPL + s < nv)
				atomicAdd(&hist[wave][bitlen(C((v[s] - mn) & E::MASK))], 1u);
		wave_sync();
		oroch_block_stats *o = out + blk;
		o->histogram[lane] = uint16_t(hist[wave][lane]);
		if (lane == 0) {
			o->histogram[64] = uint16_t(hist[wave][64]);
			o->nvalues = uint16_t(nv);
			// integer_stats.h:108-109: an empty block keeps the initial values
			o->min = nv ? uint64_t(mn) : uint64_t(E::sgn ? (E::MASK >> 1) : E::MASK);
			o->max = nv ? uint64_t(mx) : uint64_t(E::sgn ? E::SIGN : C(0));
		}
		wave_sync();
	}
}

// Seven wavefronts per SIMD is what the decoder's LDS allows for 64-bit elements;
// left alone the compiler takes registers for five (measured on the 64-bit block
// mix at 2^27 values: 0.50 ms at five, 0.48 at six, 0.47 at seven).
#ifndef DECODE_BLOCK_WAVES
#define DECODE_BLOCK_WAVES 7
#endif
#define DECODE_BLOCK_ATTR __attribute__((amdgpu_waves_per_eu(DECODE_BLOCK_WAVES, DECODE_BLOCK_WAVES)))
template <typename E>
__global__ __launch_bounds__(WAVE *WAVES_PER_WG) DECODE_BLOCK_ATTR void k_decode(const uint8_t *__restrict__ src,
								 const uint64_t *__restrict__ offsets,
								 size_t n, size_t nblocks, uint64_t stream_limit,
								 typename E::S *__restrict__ dst)
{
	using C = typename E::C;
	const int lane = lane_id();
	walk_blocks<E>(
		src, offsets, n, nblocks, stream_limit, [](size_t, const uint32_t *, uint32_t) { return WALK_DECODE; },
		[&](size_t blk, int nv, const C(&v)[VPL]) { store_block<E>(dst, blk * BLOCK, nv, lane, v); });
}

} // namespace oroch_hip

#include "decode_quad.hpp"
#include "encode_quad.hpp"
#include "select_quad.hpp"

namespace oroch_hip {

// ---- host side ----------------------------------------------------------------------

static inline size_t nblocks_of(size_t n)
{
	return (n + BLOCK - 1) / BLOCK;
}

static inline size_t align_up(size_t x, size_t a)
{
	return (x + a - 1) / a * a;
}

struct group_ws
{
	oroch_block_desc *desc;
	uint32_t *sizes;
	uint64_t *tile_sums;
	uint64_t *total;
	size_t bytes;
};

static group_ws carve_ws(void *ws, size_t n)
{
	const size_t nb = nblocks_of(n);
	const size_t ntiles = (nb + SCAN_TILE - 1) / SCAN_TILE;
	group_ws w;
	size_t off = 0;
	char *p = static_cast<char *>(ws);
	w.desc = reinterpret_cast<oroch_block_desc *>(p + off);
	off = align_up(off + nb * sizeof(oroch_block_desc), 256);
	w.sizes = reinterpret_cast<uint32_t *>(p + off);
	off = align_up(off + nb * sizeof(uint32_t), 256);
	w.tile_sums = reinterpret_cast<uint64_t *>(p + off);
	off = align_up(off + (ntiles + 1) * sizeof(uint64_t), 256);
	w.total = reinterpret_cast<uint64_t *>(p + off);
	off = align_up(off + 256, 256);
	w.bytes = off;
	return w;
}

// The *_ONE_BLOCK switches keep the one-block-per-wavefront kernels for every
// element type (the 64-bit types always use them), so that tests can compare
// the two.
static bool one_block_decode_forced()
{
	return option(OROCH_OPT_DECODE_ONE_BLOCK) != 0;
}

static bool one_block_encode_forced()
{
	return option(OROCH_OPT_ENCODE_ONE_BLOCK) != 0;
}

} // namespace oroch_hip

using namespace oroch_hip;

// Workgroups for a walk over `tiles` tiles: four tiles per wavefront, but at
// least 16384 workgroups while there is a tile for each of their wavefronts,
// and at most `most`.
static unsigned grid_for(size_t tiles, unsigned most)
{
	const size_t one_each = (tiles + WAVES_PER_WG - 1) / WAVES_PER_WG;
	const size_t low = one_each < 16384 ? one_each : 16384;
	size_t g = one_each / 4;
	g = g < low ? low : g;
	g = g < most ? g : most;
	return unsigned(g ? g : 1);
}

// Launches the selection pass: four blocks per wavefront for element types up
// to 32 bits (unless OROCH_OPT_SELECT_ONE_BLOCK is set), one block per wavefront
// otherwise.
template <typename E>
static int launch_select(const void *src, size_t n, size_t nb, oroch_block_desc *desc, uint32_t *sizes,
			 hipStream_t st)
{
	if constexpr (sizeof(typename E::C) == 4) {
		if (!option(OROCH_OPT_SELECT_ONE_BLOCK)) {
			const size_t nq = (nb + QUAD - 1) / QUAD;
			const unsigned qgrid = grid_for(nq, GRID_SELECT_QUAD);
			hipLaunchKernelGGL(k_select_quad<E>, dim3(qgrid), dim3(WAVE * WAVES_PER_WG), 0, st,
					   static_cast<const typename E::S *>(src), n, nb, desc, sizes);
			return int(hipGetLastError());
		}
	}
	const dim3 grid(grid_for(nb, GRID_SELECT)), wg(WAVE * WAVES_PER_WG);
	const typename E::S *values = static_cast<const typename E::S *>(src);
	if (desc && sizes)
		hipLaunchKernelGGL((k_select<E, true, true>), grid, wg, 0, st, values, n, nb, desc, sizes);
	else if (desc)
		hipLaunchKernelGGL((k_select<E, true, false>), grid, wg, 0, st, values, n, nb, desc, sizes);
	else if (sizes)
		hipLaunchKernelGGL((k_select<E, false, true>), grid, wg, 0, st, values, n, nb, desc, sizes);
	return int(hipGetLastError());
}

extern "C" size_t oroch_group_bound(int dtype, size_t n)
{
	if (dtype < 0 || dtype > 7)
		return 0;
	const size_t bytes = size_t(1) << (dtype >> 1);
	const size_t nb = nblocks_of(n);
	// per block at most the raw copy behind an 8-byte header, rounded to 16;
	// a constant block is 16 bytes
	return nb * align_up(8 + BLOCK * bytes, 16) + 16;
}

extern "C" size_t oroch_group_workspace(int dtype, size_t n)
{
	(void) dtype;
	return carve_ws(nullptr, n).bytes;
}

// The kernels move values with multi-dword accesses: 4-byte alignment at least
// (8 for 64-bit elements), which every allocation and every slice at a multiple
// of four elements has.
static bool values_aligned(int dtype, const void *p)
{
	const uintptr_t need = (dtype >> 1) == 3 ? 8u : 4u;
	return (reinterpret_cast<uintptr_t>(p) & (need - 1u)) == 0;
}

extern "C" int oroch_group_select(int dtype, const void *src, size_t n, oroch_block_desc *desc,
				  void *stream)
{
	if ((n && (!src || !desc)) || !values_aligned(dtype, src))
		return OROCH_E_ARGUMENT;
	const size_t nb = nblocks_of(n);
	if (nb == 0)
		return 0;
	if (nb > 0x7fffffffu * size_t(WAVES_PER_WG))
		return OROCH_E_ARGUMENT;
	hipStream_t st = static_cast<hipStream_t>(stream);
	return dispatch_dtype(dtype, [&](auto e) {
		using E = decltype(e);
		profile_scope ps(OROCH_K_SELECT, st);
		return launch_select<E>(src, n, nb, desc, nullptr, st);
	});
}

extern "C" int oroch_group_stats(int dtype, const void *src, size_t n, oroch_block_stats *stats,
				 void *stream)
{
	if ((n && (!src || !stats)) || !values_aligned(dtype, src))
		return OROCH_E_ARGUMENT;
	const size_t nb = nblocks_of(n);
	if (nb == 0)
		return 0;
	if (nb > 0x7fffffffu * size_t(WAVES_PER_WG))
		return OROCH_E_ARGUMENT;
	hipStream_t st = static_cast<hipStream_t>(stream);
	return dispatch_dtype(dtype, [&](auto e) {
		using E = decltype(e);
		profile_scope ps(OROCH_K_SELECT, st);
		hipLaunchKernelGGL(k_stats<E>, dim3(grid_for(nb, GRID_SELECT)), dim3(WAVE * WAVES_PER_WG),
				   0, st, static_cast<const typename E::S *>(src), n, nb, stats);
		return int(hipGetLastError());
	});
}

static int group_encode_impl(int dtype, const void *src, size_t n, void *dst, size_t dst_capacity,
			     uint64_t *offsets, void *workspace, size_t workspace_bytes,
			     uint64_t *total_host, void *stream)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (!offsets || (n && (!src || !dst || !workspace)))
		return OROCH_E_ARGUMENT;
	if ((reinterpret_cast<uintptr_t>(dst) & 15u) || (reinterpret_cast<uintptr_t>(workspace) & 15u) ||
	    !values_aligned(dtype, src))
		return OROCH_E_ARGUMENT;
	if (dst_capacity < oroch_group_bound(dtype, n) || workspace_bytes < oroch_group_workspace(dtype, n))
		return OROCH_E_CAPACITY;
	hipStream_t st = static_cast<hipStream_t>(stream);
	const size_t nb = nblocks_of(n);
	if (nb > 0x7fffffffu * size_t(WAVES_PER_WG))
		return OROCH_E_ARGUMENT;
	int rc = 0;
	if (nb == 0) {
		rc = int(hipMemsetAsync(offsets, 0, sizeof(uint64_t), st));
		if (rc == 0 && total_host) {
			*total_host = 0;
			rc = int(hipStreamSynchronize(st));
		}
		return rc;
	}
	group_ws w = carve_ws(workspace, n);
	// select -> scan -> encode as separate launches.  (Kernels that did all three
	// in one launch and read the values once were built and measured over two
	// rounds -- one, four and eight blocks per wavefront, offsets exchanged
	// through chained scans or a table of rows -- and every one of them was
	// slower than this: the exchange is a barrier across the chip per round of
	// tiles, and what hides it, LDS for several finished tiles per wavefront,
	// costs the occupancy the instruction streams need.)
	rc = dispatch_dtype(dtype, [&](auto e) {
			using E = decltype(e);
			int r;
			{
				profile_scope ps(OROCH_K_SELECT, st);
				r = launch_select<E>(src, n, nb, w.desc, w.sizes, st);
			}
			if (r)
				return r;
			{
				profile_scope ps(OROCH_K_SCAN, st);
				r = launch_scan<uint32_t, 0>(w.sizes, nb, w.tile_sums, w.total, offsets, st);
			}
			if (r)
				return r;
			profile_scope ps(OROCH_K_ENCODE, st);
			if constexpr (sizeof(typename E::C) == 4) {
				if (!one_block_encode_forced()) {
					const size_t nq = (nb + QUAD - 1) / QUAD;
					const unsigned qgrid = grid_for(nq, GRID_ENCODE_QUAD);
					hipLaunchKernelGGL(k_encode_quad<E>, dim3(qgrid), dim3(WAVE * WAVES_PER_WG), 0, st,
							   static_cast<const typename E::S *>(src), n, nb, w.desc, offsets,
							   static_cast<uint8_t *>(dst));
					return int(hipGetLastError());
				}
			}
			hipLaunchKernelGGL(k_encode<E>, dim3(grid_for(nb, GRID_ENCODE)),
					   dim3(WAVE * WAVES_PER_WG), 0, st,
					   static_cast<const typename E::S *>(src), n, nb, w.desc, offsets,
					   static_cast<uint8_t *>(dst));
			return int(hipGetLastError());
	});
	if (rc == 0 && total_host) {
		rc = int(hipMemcpyAsync(total_host, offsets + nb, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
		if (rc == 0)
			rc = int(hipStreamSynchronize(st));
	}
	return rc;
}


extern "C" int oroch_group_encode(int dtype, const void *src, size_t n, void *dst,
				  size_t dst_capacity, uint64_t *offsets, void *workspace,
				  size_t workspace_bytes, uint64_t *total_host, void *stream)
{
	return group_encode_impl(dtype, src, n, dst, dst_capacity, offsets, workspace, workspace_bytes,
				 total_host, stream);
}

// ---- batch encode without allocation -----------------------------------------------
//
// The workspace of the encoder (descriptors, sizes, scan and look-back state) in
// a device array of this code object instead of caller memory, for callers that
// cannot allocate.  Every GPU holds its own copy of the array at its own
// address: that address is looked up once per device and kept.

namespace oroch_hip {

constexpr size_t STATIC_SCRATCH_BYTES = size_t(4) << 20;
constexpr size_t STATIC_MAX_VALUES = size_t(1) << 24;
constexpr int STATIC_MAX_DEVICES = 64;

static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

static std::mutex g_scratch_lock;
static void *g_scratch_ptr[STATIC_MAX_DEVICES]; // device address of `scratch`, per device

// Looks up the current device's copy of the scratch array (first call per
// device) and returns its device address through *ptr.
int static_scratch_setup(void **ptr)
{
	int dev = -1;
	int rc = int(hipGetDevice(&dev));
	if (rc)
		return rc;
	if (dev < 0 || dev >= STATIC_MAX_DEVICES)
		return OROCH_E_ARGUMENT;
	std::lock_guard<std::mutex> guard(g_scratch_lock);
	if (!g_scratch_ptr[dev]) {
		void *p = nullptr;
		rc = int(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
		if (rc)
			return rc;
		if (!p || (reinterpret_cast<uintptr_t>(p) & 255u))
			return int(hipErrorInvalidSymbol);
		g_scratch_ptr[dev] = p;
	}
	if (ptr)
		*ptr = g_scratch_ptr[dev];
	return 0;
}

} // namespace oroch_hip

extern "C" size_t oroch_group_encode_static_max(int dtype)
{
	if (dtype < 0 || dtype > 7)
		return 0;
	static_assert(STATIC_MAX_VALUES % BLOCK == 0, "whole blocks");
	return carve_ws(nullptr, STATIC_MAX_VALUES).bytes <= STATIC_SCRATCH_BYTES ? STATIC_MAX_VALUES : 0;
}

extern "C" int oroch_group_encode_static(int dtype, const void *src, size_t n, void *dst,
					 size_t dst_capacity, uint64_t *offsets, void *stream)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (n > oroch_group_encode_static_max(dtype))
		return OROCH_E_TOO_LARGE;
	void *ws = nullptr;
	const int rc = static_scratch_setup(&ws);
	if (rc)
		return rc;
	// Nothing of an earlier call is carried over: descriptors and sizes are
	// written for every block before they are read, the look-back state is
	// cleared by the encoder itself.
	no_profile_scope quiet; // recording creates events
	return group_encode_impl(dtype, src, n, dst, dst_capacity, offsets, ws, STATIC_SCRATCH_BYTES, nullptr,
				 stream);
}

// Decodes with every block confined to the first stream_bytes bytes of src (and
// to offsets[nblocks], as always): for tables that come from a file or the
// network.
int oroch_hip::group_decode_within(int dtype, const void *src, const uint64_t *offsets, size_t n, void *dst,
				   uint64_t stream_bytes, void *stream)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (n && (!src || !dst || !offsets))
		return OROCH_E_ARGUMENT;
	if ((reinterpret_cast<uintptr_t>(src) & 15u) || !values_aligned(dtype, dst))
		return OROCH_E_ARGUMENT;
	const size_t nb = nblocks_of(n);
	if (nb == 0)
		return 0;
	if (nb > 0x7fffffffu * size_t(WAVES_PER_WG))
		return OROCH_E_ARGUMENT;
	hipStream_t st = static_cast<hipStream_t>(stream);
	return dispatch_dtype(dtype, [&](auto e) {
		using E = decltype(e);
		const unsigned grid = grid_for(nb, GRID_DECODE);
		profile_scope ps(OROCH_K_DECODE, st);
		if constexpr (sizeof(typename E::C) == 4) {
			if (!one_block_decode_forced()) {
				// four blocks per wavefront
				const size_t nq = (nb + QUAD - 1) / QUAD;
				const unsigned qgrid = grid_for(nq, GRID_DECODE_QUAD);
				hipLaunchKernelGGL(k_decode_quad<E>, dim3(qgrid), dim3(WAVE * WAVES_PER_WG), 0, st,
						   static_cast<const uint8_t *>(src), offsets, n, nb, stream_bytes,
						   static_cast<typename E::S *>(dst));
				return int(hipGetLastError());
			}
		}
		hipLaunchKernelGGL(k_decode<E>, dim3(grid), dim3(WAVE * WAVES_PER_WG), 0, st,
				   static_cast<const uint8_t *>(src), offsets, n, nb, stream_bytes,
				   static_cast<typename E::S *>(dst));
		return int(hipGetLastError());
	});
}

extern "C" int oroch_group_decode(int dtype, const void *src, const uint64_t *offsets, size_t n,
				  void *dst, void *stream)
{
	return group_decode_within(dtype, src, offsets, n, dst, ~uint64_t(0), stream);
}
