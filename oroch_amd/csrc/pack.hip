// pack.hip -- whole-sequence bit-packing with caller-supplied parameters.
//
//   bitpck_codec<T>::encode/decode   bitpck.h:227-256  (zigzag.h:85-93 fused)
//   bitfor_codec<T>::encode/decode   bitfor.h:59-69    (origin.h:46-54 fused)
//   bitpfr_codec<T>::encode/decode   bitpfr.h:95-124   (exception lists, offset.h)
//
// The output of a width-b packing is a run of independent 16-byte blocks,
// block j holding values [j*cap, (j+1)*cap), value k at bit k*b of the block
// read as a 128-bit little-endian integer (bitpck.h:73-128).  A workgroup
// stages a tile of values (encode) or of packed blocks (decode) in LDS with
// coalesced global accesses and converts out of LDS.
#include "common.hpp"
#include "scan.hpp"

namespace oroch_hip {

constexpr int PK_THREADS = 256;
constexpr int PK_TILE_VALUES = 4096;   // values staged per tile (upper bound)
constexpr int PK_MAX_BLOCKS = 1024;    // 16-byte blocks per tile (upper bound)

enum { VC_ZIGZAG = 0, VC_ORIGIN = 1 };

struct pack_geom
{
	uint32_t nbits, cap, magic;
	uint32_t tile_blocks;   // 16-byte blocks per tile
	uint32_t k0[2], k1[2];  // first / last value overlapping qword 0 / 1 of a block
};

// whole_words: tiles of a multiple of 64 values, so that a tile owns whole
// words of a one-bit-per-value map
static pack_geom make_geom(unsigned nbits, bool whole_words = false)
{
	pack_geom g;
	g.nbits = nbits;
	g.cap = 128u / nbits;
	g.magic = uint32_t(((uint64_t(1) << 32) + g.cap - 1) / g.cap);
	uint32_t tb = PK_TILE_VALUES / g.cap;
	g.tile_blocks = tb > PK_MAX_BLOCKS ? PK_MAX_BLOCKS : (tb ? tb : 1);
	if (whole_words) {
		// 64 blocks hold a multiple of 64 values whatever the capacity; a
		// capacity of 128 does with 32 of them
		if (g.tile_blocks >= 64)
			g.tile_blocks &= ~63u;
		else
			g.tile_blocks = g.cap % 64 == 0 ? g.tile_blocks : 64;
	}
	for (int w = 0; w < 2; w++) {
		g.k0[w] = (64u * w) / nbits;
		uint32_t last = (64u * w + 63u) / nbits;
		g.k1[w] = last > g.cap - 1 ? g.cap - 1 : last;
	}
	return g;
}

template <typename E, int VC>
__device__ __forceinline__ typename E::C value_encode(typename E::C raw, typename E::C origin)
{
	if constexpr (VC == VC_ZIGZAG)
		return zz<E>(raw);
	else
		return (raw - origin) & E::MASK;
}
template <typename E, int VC>
__device__ __forceinline__ typename E::C value_decode(typename E::C u, typename E::C origin)
{
	if constexpr (VC == VC_ZIGZAG)
		return unzz<E>(u & E::MASK);
	else
		return (u + origin) & E::MASK;
}

// The values of one 16-byte load, and back (raw bit patterns, zero-extended).
template <typename E>
__device__ __forceinline__ void split16(const uint4 &x, typename E::C (&v)[16 / E::BYTES])
{
	const uint32_t d[4] = {x.x, x.y, x.z, x.w};
	if constexpr (E::BYTES == 8) {
		v[0] = (uint64_t(d[1]) << 32) | d[0];
		v[1] = (uint64_t(d[3]) << 32) | d[2];
	} else {
		constexpr int PER = 4 / E::BYTES; // values per dword
#pragma unroll
		for (int i = 0; i < 4; i++)
#pragma unroll
			for (int k = 0; k < PER; k++)
				v[i * PER + k] = (d[i] >> (8 * E::BYTES * k)) & uint32_t(E::MASK);
	}
}
template <typename E>
__device__ __forceinline__ void join16(const typename E::C (&v)[16 / E::BYTES], uint4 *to)
{
	uint32_t d[4] = {0, 0, 0, 0};
	if constexpr (E::BYTES == 8) {
		d[0] = uint32_t(v[0]), d[1] = uint32_t(v[0] >> 32), d[2] = uint32_t(v[1]), d[3] = uint32_t(v[1] >> 32);
	} else {
		constexpr int PER = 4 / E::BYTES;
#pragma unroll
		for (int i = 0; i < 4; i++)
#pragma unroll
			for (int k = 0; k < PER; k++)
				d[i] |= uint32_t(v[i * PER + k]) << (8 * E::BYTES * k);
	}
	*to = make_uint4(d[0], d[1], d[2], d[3]);
}

// What the packer leaves behind for the exception lists of a patched sequence
// (bitpfr.h:75-85): per tile of the packer the number of exceptions and the
// position of the last one, and, for tiles with at most `room` of them, the
// exceptions themselves in order of arrival (position inside the tile, high
// part).
constexpr uint64_t XT_NONE = ~uint64_t(0);
struct exc_notes
{
	uint32_t *counts; // per tile
	uint64_t *last;   // per tile: position in the sequence, XT_NONE when the tile has none
	uint16_t *spos;   // room entries per tile
	uint64_t *sval;
	uint32_t room;
};

// Entries per tile of tile_values values: one for every 32 values.  Tiles with
// more exceptions than that are read again by the list pass.
static inline uint32_t exc_room(uint32_t tile_values)
{
	return tile_values / 32;
}

// One tile per workgroup.  EXC: also notes the exceptions (exc_notes), value v
// of the sequence being one when it does not fit nbits once the origin is taken
// off.  The high parts are at hand here, so the list pass need not look the
// values up again: a thread that has met exceptions takes places for them in a
// list in LDS (one counter), and once the tile is packed the list goes to the
// tile's notes as it is; the list pass puts the entries in order.
template <typename E, int VC, bool EXC = false>
__global__ __launch_bounds__(PK_THREADS) void k_pack(const typename E::S *src, size_t n,
						       size_t nblk, pack_geom g, typename E::C origin,
						       uint64_t *dst, exc_notes xn = exc_notes())
{
	using C = typename E::C;
	using S = typename E::S;
	constexpr int PK_VEC = 16 / E::BYTES; // values per 16-byte load
	constexpr int ROOM = PK_TILE_VALUES / 32; // exc_room of the largest tile
	__shared__ __attribute__((aligned(16))) S stage[PK_TILE_VALUES];
	// exceptions met in the tile; position of the highest, plus one; the list
	__shared__ uint32_t found, top;
	__shared__ uint16_t upos[EXC ? ROOM : 1];
	__shared__ C uval[EXC ? ROOM : 1];
	const C low = g.nbits >= uint32_t(E::W) ? E::MASK : C((C(1) << (g.nbits % E::W)) - 1);
	// (tiles start at multiples of 64 values: 16-byte loads are aligned when src is)
	const bool wide_ok = (reinterpret_cast<uintptr_t>(src) & 15u) == 0 && (g.tile_blocks * g.cap) % 64 == 0;
	const size_t tile = blockIdx.x;
	const size_t b0 = tile * g.tile_blocks;
	const uint32_t tb = uint32_t(nblk - b0 < g.tile_blocks ? nblk - b0 : g.tile_blocks);
	const size_t v0 = b0 * g.cap;
	const uint32_t slots = tb * g.cap;
	const uint32_t nvals = uint32_t(n - v0 < slots ? n - v0 : slots);
	const bool wide = nvals == slots && slots % PK_VEC == 0 && wide_ok;
	constexpr int ROUNDS = PK_TILE_VALUES / (PK_THREADS * PK_VEC);
	uint4 x[ROUNDS];
	if (wide) {
		// a full tile: 16 bytes per lane and load, all loads of the tile in
		// flight before the first is used
		const uint4 *from = reinterpret_cast<const uint4 *>(src + v0);
		const uint32_t nvec = slots / PK_VEC;
#pragma unroll
		for (int r = 0; r < ROUNDS; r++) {
			const uint32_t j = threadIdx.x + uint32_t(r) * PK_THREADS;
			x[r] = j < nvec ? from[j] : make_uint4(0, 0, 0, 0);
		}
	}
	if constexpr (EXC) {
		if (threadIdx.x == 0)
			found = top = 0;
		__syncthreads();
	}
	if (wide) {
		const uint32_t nvec = slots / PK_VEC;
		uint32_t nibs[ROUNDS];
		uint32_t mine = 0;
#pragma unroll
		for (int r = 0; r < ROUNDS; r++) {
			const uint32_t j = threadIdx.x + uint32_t(r) * PK_THREADS;
			C raw[PK_VEC];
			split16<E>(x[r], raw);
			uint32_t nib = 0;
#pragma unroll
			for (int k = 0; k < PK_VEC; k++) {
				const C u = value_encode<E, VC>(raw[k], origin);
				raw[k] = u & low;
				if constexpr (EXC)
					nib |= ((u >> g.nbits) != 0 ? 1u : 0u) << k;
			}
			nibs[r] = j < nvec ? nib : 0u;
			mine += uint32_t(__popc(nibs[r]));
			if (j < nvec)
				join16<E>(raw, reinterpret_cast<uint4 *>(stage) + j);
		}
		if constexpr (EXC) {
			if (mine) {
				uint32_t slot = atomicAdd(&found, mine);
				uint32_t hp = 0;
#pragma unroll
				for (int r = 0; r < ROUNDS; r++) {
					if (nibs[r]) {
						const uint32_t j = threadIdx.x + uint32_t(r) * PK_THREADS;
						hp = j * PK_VEC + 32u - uint32_t(__clz(nibs[r]));
						C raw[PK_VEC];
						split16<E>(x[r], raw);
#pragma unroll
						for (int k = 0; k < PK_VEC; k++) {
							if ((nibs[r] >> k) & 1u) {
								if (slot < xn.room) {
									upos[slot] = uint16_t(j * PK_VEC + uint32_t(k));
									uval[slot] = C(value_encode<E, VC>(raw[k], origin) >> g.nbits);
								}
								slot++;
							}
						}
					}
				}
				atomicMax(&top, hp);
			}
		}
	} else {
		for (uint32_t i = threadIdx.x; i < slots; i += PK_THREADS) {
			const C u = i < nvals ? value_encode<E, VC>(C(src[v0 + i]), origin) : C(0);
			stage[i] = S(u & low);
			if constexpr (EXC) {
				if ((u >> g.nbits) != 0) {
					const uint32_t slot = atomicAdd(&found, 1u);
					atomicMax(&top, i + 1u);
					if (slot < xn.room) {
						upos[slot] = uint16_t(i);
						uval[slot] = C(u >> g.nbits);
					}
				}
			}
		}
	}
	__syncthreads();
	for (uint32_t q = threadIdx.x; q < 2 * tb; q += PK_THREADS) {
		const uint32_t blk = q >> 1, w = q & 1;
		const S *vals = stage + blk * g.cap;
		// (chosen between registers: indexed by w, the launch's arguments would
		// be read from memory)
		const uint32_t k0 = w ? g.k0[1] : g.k0[0], k1 = w ? g.k1[1] : g.k1[0];
		uint64_t acc = 0;
		for (uint32_t k = k0; k <= k1; k++) {
			const uint64_t val = vals[k];
			const int pos = int(k * g.nbits) - int(64 * w);
			acc |= pos >= 0 ? val << pos : val >> (-pos);
		}
		dst[b0 * 2 + q] = acc;
	}
	if constexpr (EXC) {
		const uint32_t count = found;
		if (threadIdx.x == 0) {
			xn.counts[tile] = count;
			xn.last[tile] = top ? v0 + top - 1 : XT_NONE;
		}
		if (count <= xn.room && threadIdx.x < count) {
			const size_t at = tile * xn.room + threadIdx.x;
			xn.spos[at] = upos[threadIdx.x];
			xn.sval[at] = uint64_t(uval[threadIdx.x]);
		}
	}
}

// The exceptions of a patched sequence as the unpacker wants them: for tile t
// of the unpacker, first[t] is the number of the first exception at or beyond
// the tile's first value and where[t] that exception's position (ntiles + 1
// entries each; the last is {nexc, anything}).
struct patch_plan
{
	const uint64_t *indices; // gaps, offset.h:55-60
	const uint64_t *values;  // high parts
	const uint64_t *first;
	const uint64_t *where;
	uint32_t nbits;
};

// One tile per workgroup; all of the tile's packed words are asked for at once.
template <typename E, int VC>
__global__ __launch_bounds__(PK_THREADS) void k_unpack(const uint4 *src, size_t n, size_t nblk,
							 pack_geom g, typename E::C origin,
							 typename E::S *dst)
{
	using C = typename E::C;
	using S = typename E::S;
	constexpr int PK_VEC = 16 / E::BYTES; // values per 16-byte store
	constexpr int ROUNDS = PK_MAX_BLOCKS / PK_THREADS;
	// LDS by the launch (unpack_lds_bytes): the tile's packed words
	extern __shared__ __attribute__((aligned(16))) uint32_t unpack_lds[];
	uint32_t *stage = unpack_lds;
	// (tiles start at multiples of 64 values: 16-byte stores are aligned when dst is)
	const bool wide_ok = (reinterpret_cast<uintptr_t>(dst) & 15u) == 0 && (g.tile_blocks * g.cap) % 64 == 0;
	const size_t tile = blockIdx.x;
	const size_t b0 = tile * g.tile_blocks;
	const uint32_t tb = uint32_t(nblk - b0 < g.tile_blocks ? nblk - b0 : g.tile_blocks);
	const size_t v0 = b0 * g.cap;
	const uint32_t slots = tb * g.cap;
	const uint32_t nvals = uint32_t(n - v0 < slots ? n - v0 : slots);
	uint4 w[ROUNDS];
#pragma unroll
	for (int r = 0; r < ROUNDS; r++) {
		const uint32_t i = threadIdx.x + uint32_t(r) * PK_THREADS;
		w[r] = i < tb ? src[b0 + i] : make_uint4(0, 0, 0, 0);
	}
#pragma unroll
	for (int r = 0; r < ROUNDS; r++) {
		const uint32_t i = threadIdx.x + uint32_t(r) * PK_THREADS;
		if (i < tb)
			reinterpret_cast<uint4 *>(stage)[i] = w[r];
	}
	__syncthreads();
	if (nvals == slots && slots % PK_VEC == 0 && wide_ok) {
		// a full tile: every lane makes 16 bytes of consecutive values per store
		uint4 *to = reinterpret_cast<uint4 *>(dst + v0);
		for (uint32_t j = threadIdx.x; j < slots / PK_VEC; j += PK_THREADS) {
			uint32_t blk = __umulhi(j * PK_VEC, g.magic);
			uint32_t k = j * PK_VEC - blk * g.cap;
			C v[PK_VEC];
#pragma unroll
			for (int t = 0; t < PK_VEC; t++) {
				const C u = lds_get_bits<C>(stage, blk * 128u + k * g.nbits, g.nbits);
				v[t] = value_decode<E, VC>(u, origin) & E::MASK;
				if (++k == g.cap) {
					k = 0;
					blk++;
				}
			}
			join16<E>(v, to + j);
		}
	} else {
		for (uint32_t i = threadIdx.x; i < nvals; i += PK_THREADS) {
			const uint32_t blk = __umulhi(i, g.magic);
			const uint32_t k = i - blk * g.cap;
			const C u = lds_get_bits<C>(stage, blk * 128u + k * g.nbits, g.nbits);
			dst[v0 + i] = S(value_decode<E, VC>(u, origin));
		}
	}
}

// bitpfr.h:101-124 in one pass, one tile per workgroup.  While the tile's
// packed words arrive, the high parts of the tile's exceptions (few) go to
// their values' places in an array of LDS and are marked in a map with one bit
// per value; every value whose bit is set takes its high part from there as it
// is unpacked.
template <typename E>
__global__ __launch_bounds__(PK_THREADS) void k_unpack_patch(const uint4 *src, size_t n, size_t nblk,
							       pack_geom g, typename E::C origin,
							       typename E::S *dst, patch_plan plan)
{
	using C = typename E::C;
	using S = typename E::S;
	constexpr int PK_VEC = 16 / E::BYTES; // values per 16-byte store
	constexpr int ROUNDS = PK_MAX_BLOCKS / PK_THREADS;
	constexpr int NW = PK_TILE_VALUES / 32;
	static_assert(NW == 2 * WAVE && 32 % PK_VEC == 0, "two words of the map per lane of the first wavefront");
	// LDS by the launch (unpack_lds_bytes): the tile's packed words, then one
	// slot per value of the tile
	extern __shared__ __attribute__((aligned(16))) uint32_t unpack_lds[];
	uint32_t *stage = unpack_lds;
	C *high = reinterpret_cast<C *>(unpack_lds + g.tile_blocks * 4 + 4);
	__shared__ uint32_t marked[NW];
	__shared__ uint64_t scan_lds[PK_THREADS / WAVE];
	const size_t tile = blockIdx.x;
	const size_t b0 = tile * g.tile_blocks;
	const uint32_t tb = uint32_t(nblk - b0 < g.tile_blocks ? nblk - b0 : g.tile_blocks);
	const size_t v0 = b0 * g.cap;
	const uint32_t slots = tb * g.cap;
	const uint32_t nvals = uint32_t(n - v0 < slots ? n - v0 : slots);
	const bool wide_ok = (reinterpret_cast<uintptr_t>(dst) & 15u) == 0 && (g.tile_blocks * g.cap) % 64 == 0;
	// everything the tile needs from memory is asked for at once
	uint4 w[ROUNDS];
#pragma unroll
	for (int r = 0; r < ROUNDS; r++) {
		const uint32_t i = threadIdx.x + uint32_t(r) * PK_THREADS;
		w[r] = i < tb ? src[b0 + i] : make_uint4(0, 0, 0, 0);
	}
	// exceptions lo .. hi - 1 are the tile's, exception lo sits at value `first`
	const uint64_t lo = plan.first[tile], hi = plan.first[tile + 1], first = plan.where[tile];
	const bool few = hi - lo <= uint64_t(WAVE);
	// A step of 2^16 or more leads out of the tile whatever it is: cut there,
	// so that positions inside the tile fit 32 bits.
	auto step_to = [&](uint64_t e, uint64_t from) {
		const uint64_t gap = e < hi && e > from ? plan.indices[e] : 0;
		return e < hi && e > from ? uint32_t(gap < 0xffffu ? gap + 1 : 0x10000u) : 0u;
	};
	auto place = [&](uint64_t e, uint32_t rel, C part) {
		if (e < hi && rel < slots) {
			atomicOr(&marked[rel / 32], 1u << (rel % 32));
			high[rel] = part;
		}
	};
	if (threadIdx.x < WAVE) {
		marked[2 * threadIdx.x] = 0;
		marked[2 * threadIdx.x + 1] = 0;
		if (hi > lo && few && first - v0 < uint64_t(slots)) {
			// the usual case: the first wavefront places them, one per lane (its
			// own accesses to LDS are in order)
			const uint64_t e = lo + threadIdx.x;
			const uint32_t step = step_to(e, lo);
			const C part = e < hi ? C(plan.values[e]) << plan.nbits : C(0);
			const uint32_t rel = uint32_t(first - v0) + wave_scan_sum(step, int(threadIdx.x));
			wave_sync();
			place(e, rel, part);
		}
	}
#pragma unroll
	for (int r = 0; r < ROUNDS; r++) {
		const uint32_t i = threadIdx.x + uint32_t(r) * PK_THREADS;
		if (i < tb)
			reinterpret_cast<uint4 *>(stage)[i] = w[r];
	}
	if (hi > lo && !few && first - v0 < uint64_t(slots)) {
		__syncthreads(); // the map is clear
		uint32_t at = uint32_t(first - v0); // of exception c
		for (uint64_t c = lo; c < hi && at < slots; c += PK_THREADS) {
			const uint64_t e = c + threadIdx.x;
			const uint32_t step = step_to(e, c);
			const C part = e < hi ? C(plan.values[e]) << plan.nbits : C(0);
			uint64_t all;
			const uint64_t rel = at + wg_scan_u64(step, scan_lds, &all);
			place(e, uint32_t(rel < 0x7fffffffu ? rel : 0x7fffffffu), part);
			// exception c + PK_THREADS sits one more step on
			const uint64_t next = at + all + step_to(c + PK_THREADS, c);
			at = uint32_t(next < 0x7fffffffu ? next : 0x7fffffffu);
		}
	}
	__syncthreads();
	if (nvals == slots && slots % PK_VEC == 0 && wide_ok) {
		// a full tile: every lane makes 16 bytes of consecutive values per store
		uint4 *to = reinterpret_cast<uint4 *>(dst + v0);
		for (uint32_t j = threadIdx.x; j < slots / PK_VEC; j += PK_THREADS) {
			uint32_t blk = __umulhi(j * PK_VEC, g.magic);
			uint32_t k = j * PK_VEC - blk * g.cap;
			const uint32_t mine = marked[(j * PK_VEC) / 32] >> ((j * PK_VEC) % 32);
			C v[PK_VEC];
#pragma unroll
			for (int t = 0; t < PK_VEC; t++) {
				C u = lds_get_bits<C>(stage, blk * 128u + k * g.nbits, g.nbits);
				if ((mine >> t) & 1u)
					u |= high[j * PK_VEC + t];
				v[t] = value_decode<E, VC_ORIGIN>(u, origin) & E::MASK;
				if (++k == g.cap) {
					k = 0;
					blk++;
				}
			}
			join16<E>(v, to + j);
		}
	} else {
		for (uint32_t i = threadIdx.x; i < nvals; i += PK_THREADS) {
			const uint32_t blk = __umulhi(i, g.magic);
			const uint32_t k = i - blk * g.cap;
			C u = lds_get_bits<C>(stage, blk * 128u + k * g.nbits, g.nbits);
			if ((marked[i / 32] >> (i % 32)) & 1u)
				u |= high[i];
			dst[v0 + i] = S(value_decode<E, VC_ORIGIN>(u, origin));
		}
	}
}

// Where the unpacker's tiles find their exceptions (see patch_plan): tile of
// SCAN_TILE exceptions per workgroup; an exception that is the first at or
// beyond the start of some tiles writes their entries, the last exception also
// those of the tiles behind it.
__global__ __launch_bounds__(SCAN_THREADS) void k_patch_plan(const uint64_t *indices, size_t nexc,
							      const uint64_t *tile_base, uint64_t tile_vals,
							      size_t ntiles, uint64_t *first, uint64_t *where)
{
	__shared__ uint64_t lds[SCAN_THREADS / WAVE];
	const size_t base = size_t(blockIdx.x) * SCAN_TILE + size_t(threadIdx.x) * SCAN_ITEMS;
	uint64_t g[SCAN_ITEMS];
	uint64_t s = 0;
#pragma unroll
	for (int i = 0; i < SCAN_ITEMS; i++) {
		g[i] = base + i < nexc ? indices[base + i] + 1 : 0;
		s += g[i];
	}
	uint64_t all;
	uint64_t run = tile_base[blockIdx.x] + wg_scan_u64(s, lds, &all) - s;
	auto fill = [&](uint64_t from, uint64_t to, uint64_t e, uint64_t pos) {
		// entries from .. to (inclusive); long runs (sparse exceptions) are rare
		for (uint64_t t = from; t <= to && t <= ntiles; t++) {
			first[t] = e;
			where[t] = pos;
		}
	};
#pragma unroll
	for (int i = 0; i < SCAN_ITEMS; i++) {
		if (base + i < nexc) {
			const uint64_t pos = run + g[i] - 1;
			const uint64_t tile = pos / tile_vals;
			// tiles after the one of the exception before me, up to mine
			const uint64_t after = base + i == 0 ? 0 : (run - 1) / tile_vals + 1;
			fill(after, tile, base + i, pos);
			if (base + i + 1 == nexc)
				fill(tile + 1, ntiles, nexc, 0);
		}
		run += g[i];
	}
}

// Random access, bitpck.h:203-225, 258-266 and bitfor.h:71-74: value number
// index[i] of a packed sequence, straight from global memory (each value
// touches one or two qwords of its 16-byte block).
template <typename E, int VC>
__global__ __launch_bounds__(PK_THREADS) void k_fetch(const uint64_t *src, size_t n,
							const uint64_t *index, size_t m, pack_geom g,
							typename E::C origin, typename E::S *dst)
{
	using C = typename E::C;
	using S = typename E::S;
	const C low = g.nbits >= uint32_t(E::W) ? E::MASK : C((C(1) << (g.nbits % E::W)) - 1);
	for (size_t i = size_t(blockIdx.x) * PK_THREADS + threadIdx.x; i < m; i += size_t(gridDim.x) * PK_THREADS) {
		const uint64_t at = index[i];
		C u = 0;
		if (at < n) {
			const uint64_t blk = at / g.cap;
			const uint32_t bit = uint32_t(at - blk * g.cap) * g.nbits; // 0..127
			const uint64_t *q = src + blk * 2;
			const uint64_t lo = q[bit >> 6];
			const uint32_t sh = bit & 63u;
			uint64_t x = lo >> sh;
			if (sh + g.nbits > 64u)
				x |= q[1] << (64u - sh);
			u = C(x) & low;
		}
		dst[i] = S(value_decode<E, VC>(u, origin));
	}
}

// ---- bitpfr exception lists ---------------------------------------------------------
//
// The packer has noted per tile how many exceptions it holds, where the last
// one sits and (unless they are too many) the exceptions themselves, in no
// particular order; a scan of the counts gives every tile the rank of its
// first exception.  One wavefront
// per tile then writes the tile's part of the two lists: gap-coded positions
// (offset.h:55-60 with offset 1) and high parts (bitpfr.h:75-85).

constexpr int XL_THREADS = 256;
constexpr int XL_MAP = PK_TILE_VALUES / 32;   // words of a map with one bit per value of a tile
constexpr int XL_ROOM = PK_TILE_VALUES / 32;  // exc_room of the largest tile

template <typename E>
__global__ __launch_bounds__(XL_THREADS) void k_exc_lists(const typename E::S *src, size_t n,
							    typename E::C origin, uint32_t nbits,
							    uint32_t tile_values, size_t ntiles, exc_notes xn,
							    const uint64_t *rank_base, uint64_t *indices,
							    uint64_t *values, size_t capacity, uint64_t *nexc)
{
	using C = typename E::C;
	__shared__ uint32_t maps[XL_THREADS / WAVE][XL_MAP], unders[XL_THREADS / WAVE][XL_MAP];
	__shared__ uint16_t sorteds[XL_THREADS / WAVE][XL_ROOM];
	const int lane = lane_id();
	const size_t tile = size_t(blockIdx.x) * (XL_THREADS / WAVE) + threadIdx.x / WAVE;
	if (tile >= ntiles)
		return;
	if (tile + 1 == ntiles && lane == 0)
		*nexc = rank_base[ntiles];
	// Everything the tile may need from memory is asked for at once (one wait
	// instead of a chain of them): its count and rank, the last positions of the
	// 64 tiles before it, and its first notes, whether they turn out to be
	// wanted or not (what lies past the count is never looked at).
	const uint32_t count_l = xn.counts[tile];
	const uint64_t rank0 = rank_base[tile];
	const size_t lo0 = tile > WAVE ? tile - WAVE : 0;
	const uint64_t near = lo0 + size_t(lane) < tile ? xn.last[lo0 + size_t(lane)] : XT_NONE;
	const uint16_t *spos = xn.spos + tile * xn.room;
	const uint64_t *sval = xn.sval + tile * xn.room;
	const uint32_t e1 = uint32_t(lane) + WAVE < xn.room ? uint32_t(lane) + WAVE : xn.room - 1u;
	uint32_t p[2] = {spos[lane], spos[e1]}, r[2];
	uint64_t hv[2] = {sval[lane], sval[e1]};
	const uint32_t count = uniform(count_l);
	if (count == 0)
		return;
	// The last exception before this tile: the nearest earlier tile that has
	// one, 64 tiles per step.
	uint64_t before = XT_NONE;
	{
		const uint64_t have = __ballot(near != XT_NONE);
		if (have)
			before = read_lane(near, 63 - __builtin_clzll(have));
	}
	for (size_t hi = lo0; hi > 0 && before == XT_NONE;) {
		const size_t lo = hi > WAVE ? hi - WAVE : 0;
		const size_t t = lo + size_t(lane);
		const uint64_t x = t < hi ? xn.last[t] : XT_NONE;
		const uint64_t have = __ballot(x != XT_NONE);
		if (have)
			before = read_lane(x, 63 - __builtin_clzll(have));
		hi = lo;
	}
	const uint64_t v0 = uint64_t(tile) * tile_values;
	// the position a gap counts from: one past the exception before, 0 at the start
	const uint64_t next0 = before == XT_NONE ? 0 : before + 1;
	if (count <= xn.room) {
		// The packer's notes, in the order its threads met the exceptions: a map
		// of the positions gives every entry its rank in the tile.
		static_assert(XL_ROOM == 2 * WAVE && XL_MAP == 2 * WAVE, "two entries and two words of the map per lane");
		uint32_t *map = maps[threadIdx.x / WAVE], *under = unders[threadIdx.x / WAVE];
		uint16_t *sorted = sorteds[threadIdx.x / WAVE];
		map[2 * lane] = map[2 * lane + 1] = 0;
		wave_sync();
#pragma unroll
		for (int t = 0; t < 2; t++) {
			const uint32_t e = uint32_t(lane) + uint32_t(t) * WAVE;
			if (e < count)
				atomicOr(&map[p[t] / 32], 1u << (p[t] % 32));
		}
		wave_sync();
		{
			const uint32_t w0 = map[2 * lane], w1 = map[2 * lane + 1];
			const uint32_t c0 = uint32_t(__popc(w0)), c = c0 + uint32_t(__popc(w1));
			const uint32_t incl = wave_scan_sum(c, lane);
			under[2 * lane] = incl - c;
			under[2 * lane + 1] = incl - c + c0;
		}
		wave_sync();
#pragma unroll
		for (int t = 0; t < 2; t++) {
			const uint32_t e = uint32_t(lane) + uint32_t(t) * WAVE;
			if (e < count) {
				r[t] = under[p[t] / 32] + uint32_t(__popc(map[p[t] / 32] & ((1u << (p[t] % 32)) - 1u)));
				sorted[r[t]] = uint16_t(p[t]);
			}
		}
		wave_sync();
#pragma unroll
		for (int t = 0; t < 2; t++) {
			const uint32_t e = uint32_t(lane) + uint32_t(t) * WAVE;
			if (e < count && rank0 + r[t] < capacity) {
				const uint64_t from = r[t] ? v0 + sorted[r[t] - 1] + 1 : next0;
				indices[rank0 + r[t]] = v0 + p[t] - from;
				values[rank0 + r[t]] = hv[t];
			}
		}
		return;
	}
	// a tile with many exceptions: its values once more, 64 at a time
	const uint64_t end = v0 + tile_values < n ? v0 + tile_values : n;
	uint64_t rank = rank0, next = next0;
	for (uint64_t base = v0; base < end; base += WAVE) {
		const uint64_t pos = base + uint64_t(lane);
		const C u = pos < end ? C((C(src[pos]) - origin) & E::MASK) : C(0);
		const bool mine = (u >> nbits) != 0;
		const uint64_t all = __ballot(mine);
		if (all == 0)
			continue;
		const uint64_t lower = all & ((uint64_t(1) << lane) - 1u);
		const uint64_t r = rank + uint64_t(__popcll(lower));
		const uint64_t from = lower ? base + uint64_t(63 - __builtin_clzll(lower)) + 1 : next;
		if (mine && r < capacity) {
			indices[r] = pos - from;
			values[r] = uint64_t(u >> nbits);
		}
		rank += uint64_t(__popcll(all));
		next = base + uint64_t(63 - __builtin_clzll(all)) + 1;
	}
}

// Patch pass of decode, bitpfr.h:114-124: tile t of 2048 exceptions.
template <typename E>
__global__ __launch_bounds__(SCAN_THREADS) void k_patch(const uint64_t *indices,
							  const uint64_t *values, size_t nexc,
							  const uint64_t *tile_base,
							  typename E::C origin, uint32_t nbits, size_t n,
							  typename E::S *dst)
{
	using C = typename E::C;
	using S = typename E::S;
	__shared__ uint64_t lds[SCAN_THREADS / WAVE];
	const size_t base = size_t(blockIdx.x) * SCAN_TILE + size_t(threadIdx.x) * SCAN_ITEMS;
	uint64_t g[SCAN_ITEMS];
	uint64_t s = 0;
#pragma unroll
	for (int i = 0; i < SCAN_ITEMS; i++) {
		g[i] = base + i < nexc ? indices[base + i] + 1 : 0;
		s += g[i];
	}
	uint64_t all;
	uint64_t run = tile_base[blockIdx.x] + wg_scan_u64(s, lds, &all) - s;
#pragma unroll
	for (int i = 0; i < SCAN_ITEMS; i++) {
		if (base + i < nexc) {
			const uint64_t pos = run + g[i] - 1;
			if (pos < n) {
				C u = (C(dst[pos]) - origin) & E::MASK;
				u |= C(values[base + i]) << nbits;
				dst[pos] = S((u + origin) & E::MASK);
			}
		}
		run += g[i];
	}
}

// Dynamic LDS of k_unpack: the packed words of a tile (+16 bytes that a field
// read may touch), with patches one slot of the compute type per value.
static size_t unpack_lds_bytes(const pack_geom &g, int dtype, bool patch)
{
	const size_t stage = (size_t(g.tile_blocks) * 4 + 4) * sizeof(uint32_t);
	const size_t slot = (dtype >> 1) == 3 ? 8 : 4;
	return stage + (patch ? size_t(g.tile_blocks) * g.cap * slot + 64 : 0);
}

template <int VC>
static int do_pack(int dtype, const void *src, size_t n, uint64_t origin, unsigned nbits, void *dst,
		   void *stream)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	const unsigned W = 8u << (dtype >> 1);
	if (nbits < 1 || nbits > W)
		return OROCH_E_NBITS;
	if (n == 0)
		return 0;
	if (!src || !dst || (reinterpret_cast<uintptr_t>(dst) & 7u))
		return OROCH_E_ARGUMENT;
	const pack_geom g = make_geom(nbits, true);
	const size_t nblk = (n + g.cap - 1) / g.cap;
	const size_t ntiles = (nblk + g.tile_blocks - 1) / g.tile_blocks;
	hipStream_t st = static_cast<hipStream_t>(stream);
	return dispatch_dtype(dtype, [&](auto e) {
		using E = decltype(e);
		profile_scope ps(OROCH_K_PACK, st);
		if (ntiles > 0x7fffffffu)
			return int(OROCH_E_ARGUMENT);
		hipLaunchKernelGGL((k_pack<E, VC>), dim3(unsigned(ntiles)), dim3(PK_THREADS), 0, st,
				   static_cast<const typename E::S *>(src), n, nblk, g,
				   typename E::C(origin & uint64_t(E::MASK)), static_cast<uint64_t *>(dst));
		return int(hipGetLastError());
	});
}

template <int VC>
static int do_unpack(int dtype, const void *src, size_t n, uint64_t origin, unsigned nbits,
		     void *dst, void *stream)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	const unsigned W = 8u << (dtype >> 1);
	if (nbits < 1 || nbits > W)
		return OROCH_E_NBITS;
	if (n == 0)
		return 0;
	if (!src || !dst || (reinterpret_cast<uintptr_t>(src) & 15u))
		return OROCH_E_ARGUMENT;
	const pack_geom g = make_geom(nbits, true);
	const size_t nblk = (n + g.cap - 1) / g.cap;
	const size_t ntiles = (nblk + g.tile_blocks - 1) / g.tile_blocks;
	hipStream_t st = static_cast<hipStream_t>(stream);
	return dispatch_dtype(dtype, [&](auto e) {
		using E = decltype(e);
		profile_scope ps(OROCH_K_UNPACK, st);
		if (ntiles > 0x7fffffffu)
			return int(OROCH_E_ARGUMENT);
		hipLaunchKernelGGL((k_unpack<E, VC>), dim3(unsigned(ntiles)), dim3(PK_THREADS),
				   unpack_lds_bytes(g, dtype, false), st,
				   static_cast<const uint4 *>(src), n, nblk, g,
				   typename E::C(origin & uint64_t(E::MASK)),
				   static_cast<typename E::S *>(dst));
		return int(hipGetLastError());
	});
}

// The packer's tiles hold between 2048 and 4096 values whatever the width
// (make_geom with whole words).
constexpr size_t PK_TILE_MIN = 2048;

struct pfr_ws
{
	uint32_t *counts;     // per tile of the packer
	uint64_t *last;
	uint64_t *rank_base;  // tiles + 1
	uint64_t *tile_sums;
	uint64_t *total;
	uint16_t *spos;       // the packer's notes, exc_room entries per tile
	uint64_t *sval;
	uint64_t *plan_first; // per tile of the unpacker (patch_plan)
	uint64_t *plan_where;
	size_t bytes;
};

static size_t up256(size_t x)
{
	return (x + 255) / 256 * 256;
}

static pfr_ws carve_pfr(void *ws, size_t n)
{
	const size_t ntiles = n / PK_TILE_MIN + 2;
	// the decoder scans up to n exceptions in tiles of SCAN_TILE
	const size_t stiles = scan_tiles(n > ntiles ? n : ntiles) + 1;
	pfr_ws w;
	char *p = static_cast<char *>(ws);
	size_t off = 0;
	w.counts = reinterpret_cast<uint32_t *>(p + off);
	off = up256(off + (ntiles + 1) * sizeof(uint32_t));
	w.last = reinterpret_cast<uint64_t *>(p + off);
	off = up256(off + (ntiles + 1) * sizeof(uint64_t));
	w.rank_base = reinterpret_cast<uint64_t *>(p + off);
	off = up256(off + (ntiles + 2) * sizeof(uint64_t));
	w.tile_sums = reinterpret_cast<uint64_t *>(p + off);
	off = up256(off + (stiles + 1) * sizeof(uint64_t));
	w.total = reinterpret_cast<uint64_t *>(p + off);
	off = up256(off + 256);
	// one entry for every 32 values of a tile (exc_room), tiles of at most 4096
	// values: n / 32 entries and those of one tile more
	const size_t entries = n / 32 + 2 * exc_room(PK_TILE_VALUES);
	w.sval = reinterpret_cast<uint64_t *>(p + off);
	off = up256(off + entries * sizeof(uint64_t));
	w.spos = reinterpret_cast<uint16_t *>(p + off);
	off = up256(off + entries * sizeof(uint16_t));
	// the unpacker's tiles hold 1024 values at least
	const size_t utiles = n / 1024 + 3;
	w.plan_first = reinterpret_cast<uint64_t *>(p + off);
	off = up256(off + utiles * sizeof(uint64_t));
	w.plan_where = reinterpret_cast<uint64_t *>(p + off);
	off = up256(off + utiles * sizeof(uint64_t));
	w.bytes = off;
	return w;
}

} // namespace oroch_hip

using namespace oroch_hip;

extern "C" size_t oroch_bitpck_capacity(unsigned nbits)
{
	return nbits ? 128u / nbits : 0;
}

extern "C" size_t oroch_bitpck_space(size_t nvalues, unsigned nbits)
{
	if (!nbits || !nvalues)
		return 0;
	const size_t c = 128u / nbits;
	return 16 * ((nvalues + c - 1) / c);
}

extern "C" int oroch_bitpck_encode(int dtype, const void *src, size_t n, unsigned nbits, void *dst,
				   void *stream)
{
	return do_pack<VC_ZIGZAG>(dtype, src, n, 0, nbits, dst, stream);
}

extern "C" int oroch_bitpck_decode(int dtype, const void *src, size_t n, unsigned nbits, void *dst,
				   void *stream)
{
	return do_unpack<VC_ZIGZAG>(dtype, src, n, 0, nbits, dst, stream);
}

extern "C" int oroch_bitfor_encode(int dtype, const void *src, size_t n, uint64_t origin,
				   unsigned nbits, void *dst, void *stream)
{
	return do_pack<VC_ORIGIN>(dtype, src, n, origin, nbits, dst, stream);
}

extern "C" int oroch_bitfor_decode(int dtype, const void *src, size_t n, uint64_t origin,
				   unsigned nbits, void *dst, void *stream)
{
	return do_unpack<VC_ORIGIN>(dtype, src, n, origin, nbits, dst, stream);
}

extern "C" int oroch_bitpck_fetch(int dtype, int vcodec, const void *src, size_t n, uint64_t origin,
				  unsigned nbits, const uint64_t *index, size_t m, void *dst, void *stream)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	const unsigned W = 8u << (dtype >> 1);
	if (nbits < 1 || nbits > W)
		return OROCH_E_NBITS;
	if (vcodec != VC_ZIGZAG && vcodec != VC_ORIGIN)
		return OROCH_E_ARGUMENT;
	if (m == 0)
		return 0;
	if (!src || !index || !dst || (reinterpret_cast<uintptr_t>(src) & 7u))
		return OROCH_E_ARGUMENT;
	const pack_geom g = make_geom(nbits);
	hipStream_t st = static_cast<hipStream_t>(stream);
	const size_t wgs = (m + PK_THREADS - 1) / PK_THREADS;
	const unsigned grid = unsigned(wgs < 65536 ? wgs : 65536);
	return dispatch_dtype(dtype, [&](auto e) {
		using E = decltype(e);
		profile_scope ps(OROCH_K_UNPACK, st);
		const typename E::C org = typename E::C(origin & uint64_t(E::MASK));
		if (vcodec == VC_ZIGZAG)
			hipLaunchKernelGGL((k_fetch<E, VC_ZIGZAG>), dim3(grid), dim3(PK_THREADS), 0, st,
					   static_cast<const uint64_t *>(src), n, index, m, g, org,
					   static_cast<typename E::S *>(dst));
		else
			hipLaunchKernelGGL((k_fetch<E, VC_ORIGIN>), dim3(grid), dim3(PK_THREADS), 0, st,
					   static_cast<const uint64_t *>(src), n, index, m, g, org,
					   static_cast<typename E::S *>(dst));
		return int(hipGetLastError());
	});
}

extern "C" size_t oroch_bitpfr_workspace(size_t n)
{
	return carve_pfr(nullptr, n).bytes;
}

extern "C" int oroch_bitpfr_encode(int dtype, const void *src, size_t n, uint64_t origin,
				   unsigned nbits, void *dst, uint64_t *indices, uint64_t *values,
				   size_t exc_capacity, uint64_t *nexc, void *workspace,
				   size_t workspace_bytes, void *stream)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	const unsigned W = 8u << (dtype >> 1);
	if (nbits < 1 || nbits >= W)
		return OROCH_E_NBITS; // bitpfr.h:65: the mask is (1 << nbits) - 1, nbits < bits(T)
	if (!nexc || !workspace || (exc_capacity && (!indices || !values)))
		return OROCH_E_ARGUMENT;
	if (workspace_bytes < oroch_bitpfr_workspace(n))
		return OROCH_E_CAPACITY;
	hipStream_t st = static_cast<hipStream_t>(stream);
	if (n == 0)
		return int(hipMemsetAsync(nexc, 0, sizeof(uint64_t), st));
	if (!src || !dst || (reinterpret_cast<uintptr_t>(dst) & 7u))
		return OROCH_E_ARGUMENT;
	pfr_ws w = carve_pfr(workspace, n);
	// One pass over the values packs them and notes the exceptions tile by
	// tile; a scan of the tiles' counts places them in the lists.
	const pack_geom g = make_geom(nbits, true);
	const size_t nblk = (n + g.cap - 1) / g.cap;
	const size_t ptiles = (nblk + g.tile_blocks - 1) / g.tile_blocks;
	const uint32_t tile_values = g.tile_blocks * g.cap;
	if (tile_values < PK_TILE_MIN || tile_values > uint32_t(PK_TILE_VALUES) || tile_values % 64 || ptiles > 0x7fffffffu)
		return OROCH_E_ARGUMENT;
	const exc_notes xn = {w.counts, w.last, w.spos, w.sval, exc_room(tile_values)};
	return dispatch_dtype(dtype, [&](auto e) {
		using E = decltype(e);
		using C = typename E::C;
		const typename E::S *s = static_cast<const typename E::S *>(src);
		const C org = C(origin & uint64_t(E::MASK));
		{
			profile_scope ps(OROCH_K_PACK, st);
			hipLaunchKernelGGL((k_pack<E, VC_ORIGIN, true>), dim3(unsigned(ptiles)), dim3(PK_THREADS), 0,
					   st, s, n, nblk, g, org, static_cast<uint64_t *>(dst), xn);
		}
		profile_scope ps(OROCH_K_EXCEPT, st);
		int r = launch_scan<uint32_t, 0>(w.counts, ptiles, w.tile_sums, w.total, w.rank_base, st);
		if (r)
			return r;
		constexpr int PER = XL_THREADS / WAVE;
		hipLaunchKernelGGL(k_exc_lists<E>, dim3(unsigned((ptiles + PER - 1) / PER)), dim3(XL_THREADS), 0, st, s,
				   n, org, nbits, tile_values, ptiles, xn, w.rank_base, indices, values, exc_capacity,
				   nexc);
		return int(hipGetLastError());
	});
}

extern "C" int oroch_bitpfr_patch(int dtype, void *dst, size_t n, uint64_t origin, unsigned nbits,
				  const uint64_t *indices, const uint64_t *values, size_t nexc,
				  void *workspace, size_t workspace_bytes, void *stream)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	const unsigned W = 8u << (dtype >> 1);
	if (nbits < 1 || nbits >= W)
		return OROCH_E_NBITS;
	if (nexc == 0)
		return 0;
	if (nexc > n || !dst || !indices || !values || !workspace)
		return OROCH_E_ARGUMENT;
	if (workspace_bytes < oroch_bitpfr_workspace(n))
		return OROCH_E_CAPACITY;
	hipStream_t st = static_cast<hipStream_t>(stream);
	pfr_ws w = carve_pfr(workspace, n);
	const size_t ntiles = scan_tiles(nexc);
	return dispatch_dtype(dtype, [&](auto e) {
		using E = decltype(e);
		using C = typename E::C;
		profile_scope ps(OROCH_K_EXCEPT, st);
		hipLaunchKernelGGL((k_tile_sums<uint64_t, 1>), dim3(unsigned(ntiles)), dim3(SCAN_THREADS), 0,
				   st, indices, nexc, w.tile_sums);
		hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(SCAN_THREADS), 0, st, w.tile_sums, ntiles,
				   w.total);
		hipLaunchKernelGGL(k_patch<E>, dim3(unsigned(ntiles)), dim3(SCAN_THREADS), 0, st, indices,
				   values, nexc, w.tile_sums, C(origin & uint64_t(E::MASK)), nbits, n,
				   static_cast<typename E::S *>(dst));
		return int(hipGetLastError());
	});
}

extern "C" int oroch_bitpfr_decode(int dtype, const void *src, size_t n, uint64_t origin,
				   unsigned nbits, const uint64_t *indices, const uint64_t *values,
				   size_t nexc, void *dst, void *workspace, size_t workspace_bytes,
				   void *stream)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	const unsigned W = 8u << (dtype >> 1);
	if (nbits < 1 || nbits >= W)
		return OROCH_E_NBITS;
	if (nexc > n || (nexc && (!indices || !values || !workspace)))
		return OROCH_E_ARGUMENT;
	if (nexc && workspace_bytes < oroch_bitpfr_workspace(n))
		return OROCH_E_CAPACITY;
	if (nexc == 0 || n == 0)
		return do_unpack<VC_ORIGIN>(dtype, src, n, origin, nbits, dst, stream);
	if (!src || !dst || (reinterpret_cast<uintptr_t>(src) & 15u))
		return OROCH_E_ARGUMENT;
	// One pass over the packed words: the exceptions are sorted onto the
	// unpacker's tiles first (a scan of their gaps), and each tile patches its
	// own values while their packed words are still in LDS.
	hipStream_t st = static_cast<hipStream_t>(stream);
	pfr_ws w = carve_pfr(workspace, n);
	const pack_geom g = make_geom(nbits, true);
	const size_t nblk = (n + g.cap - 1) / g.cap;
	const size_t ntiles = (nblk + g.tile_blocks - 1) / g.tile_blocks;
	const uint64_t tile_vals = uint64_t(g.tile_blocks) * g.cap;
	const size_t stiles = scan_tiles(nexc);
	return dispatch_dtype(dtype, [&](auto e) {
		using E = decltype(e);
		using C = typename E::C;
		{
			profile_scope ps(OROCH_K_EXCEPT, st);
			// (entries that lists with impossible gaps leave unwritten read as "no exceptions")
			// (the two arrays are neighbours in the workspace)
			const int r = int(hipMemsetAsync(w.plan_first, 0,
							 size_t(reinterpret_cast<char *>(w.plan_where + ntiles + 1) -
								reinterpret_cast<char *>(w.plan_first)), st));
			if (r)
				return r;
			hipLaunchKernelGGL((k_tile_sums<uint64_t, 1>), dim3(unsigned(stiles)), dim3(SCAN_THREADS), 0, st,
					   indices, nexc, w.tile_sums);
			hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(SCAN_THREADS), 0, st, w.tile_sums, stiles, w.total);
			hipLaunchKernelGGL(k_patch_plan, dim3(unsigned(stiles)), dim3(SCAN_THREADS), 0, st, indices, nexc,
					   w.tile_sums, tile_vals, ntiles, w.plan_first, w.plan_where);
		}
		profile_scope ps(OROCH_K_UNPACK, st);
		const patch_plan plan = {indices, values, w.plan_first, w.plan_where, nbits};
		if (ntiles > 0x7fffffffu)
			return int(OROCH_E_ARGUMENT);
		hipLaunchKernelGGL(k_unpack_patch<E>, dim3(unsigned(ntiles)), dim3(PK_THREADS),
				   unpack_lds_bytes(g, dtype, true), st,
				   static_cast<const uint4 *>(src), n, nblk, g, C(origin & uint64_t(E::MASK)),
				   static_cast<typename E::S *>(dst), plan);
		return int(hipGetLastError());
	});
}
