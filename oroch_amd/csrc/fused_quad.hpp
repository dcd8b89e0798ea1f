// fused_quad.hpp -- selection, offsets and encoding in ONE launch, four blocks
// per wavefront: the values are read from HBM once (integer_group.h:44-60 for
// every block of the array; integer_codec.h:234-402).
//
// A tile is one quad (four consecutive blocks).  A workgroup is twelve
// wavefronts that select and encode (workers) and a thirteenth that finds out
// where the workgroup's tiles go (the exchanger); one workgroup per CU, all
// resident.  In round r the workers of workgroup b take the twelve consecutive
// tiles (r * B + b) * 12 .. + 11, B being the number of workgroups.
//
// Where a tile goes is the sum of the sizes of all tiles before it, which other
// workgroups are working out at the same time.  The sizes are exchanged through
// one small table in device memory, zero before the launch:
//
//   row[r][b]  16 bits: bit 15 set once workgroup b has the sizes of all its
//              tiles of round r, bits 14..0 their sum in units of 16 bytes;
//              written by whichever worker of the workgroup is last with its
//              size in that round.
//
// The offset of workgroup b's tiles of round r is then the sum of all complete
// rows before r, which the exchanger keeps as it goes, plus row[r][0 .. b-1]:
// one hop from the stores of the others to its own loads, the whole row (512
// bytes for 256 workgroups) in one load per lane.
//
// Nobody waits for an offset with work at hand.  A worker encodes its quad into
// a ring in LDS and goes on to the next tile; the exchanger polls the table and
// hands the offsets over through LDS; the worker copies a finished tile out
// when its offset is there, and waits only when its ring is full.  The polling
// is kept away from the workers because a wavefront's memory operations
// complete in order: a polling load, a publishing store (several microseconds
// each under load, every lane's piece a request of its own to the memory
// fabric) would hold back every later value load of a worker.  For the same
// reason the table is stored to by workers (after their value loads for the
// next round are under way) and only loaded by the exchanger: a load issued
// behind a store of the same wavefront waits for that store's acknowledgement.
//
// Every entry is written by one agent-scope store and polled with agent-scope
// loads; flag and value travel in the same word, so no fence is needed.  Waits
// are bounded: when one runs out the abort word is set, nothing more is
// written, and the host falls back to the separate-launch path.
//
// Included by group.hip after select_quad.hpp; 8-, 16- and 32-bit element types.
#pragma once

namespace oroch_hip {

#ifndef FQ_NWORKERS
#define FQ_NWORKERS 11
#endif
#ifndef FQ_EU
#define FQ_EU 3
#endif
constexpr int FQ_WORKERS = FQ_NWORKERS;   // with the exchanger three wavefronts to a SIMD: registers for that many, no spills (0.69 against 0.85 ms for twelve workers at 2^28)
constexpr int FQ_THREADS = WAVE * (FQ_WORKERS + 1);
constexpr int FQ_MAX_GROUPS = 256;        // workgroups = entries of a row (one load of 8 bytes per lane)
constexpr uint32_t FQ_ROW_VALID = 0x8000u;
#ifndef FQ_IDLE
#define FQ_IDLE 22
#endif
constexpr uint32_t FQ_IDLE_LIMIT = 1u << FQ_IDLE;

#ifdef FQ_STATS // experiment builds: counts of what the wavefronts met, printed by the host
__device__ unsigned long long fq_stats[32];
#define FQ_COUNT(i, lane) do { if ((lane) == 0) st.dbg[i] += 1; } while (0)
#define FQ_TIME(i, lane, t0) do { if ((lane) == 0) st.dbg[i] += uint32_t((__builtin_amdgcn_s_memtime() - (t0)) >> 4); } while (0)
#define FQ_NOW() __builtin_amdgcn_s_memtime()
#else
#define FQ_COUNT(i, lane) do { } while (0)
#define FQ_TIME(i, lane, t0) do { (void) (t0); } while (0)
#define FQ_NOW() 0ull
#endif

struct fq_state
{
	uint16_t *rows;       // [rounds][FQ_MAX_GROUPS]
	uint32_t *abort_flag;
	uint32_t *dbg;        // experiment builds: this wavefront's counters (LDS)
};

// Lane p, p + 64, ... packs word p of a class of blocks from the slots into
// the image of the tile (pack_quad_words with an LDS destination).
template <typename E, int NB>
__device__ __forceinline__ void fq_pack_words(const quad_class &c, const typename E::C *slots,
					      uint32_t data_at, int lane, uint8_t *img)
{
	using C = typename E::C;
	constexpr uint32_t NBLK = (BLOCK + 128 / NB - 1) / (128 / NB);
	const uint32_t P = c.count * NBLK;
	asm volatile("" : "+v"(lane));
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
			*reinterpret_cast<uint4 *>(img + at + j * 16u) = make_uint4(w[0], w[1], w[2], w[3]);
		}
	}
}

template <typename E, int LO, int HI>
__device__ __forceinline__ void fq_pack_dispatch(uint32_t nb, const quad_class &c, const typename E::C *slots,
						 uint32_t data_at, int lane, uint8_t *img)
{
	if constexpr (LO == HI) {
		fq_pack_words<E, LO>(c, slots, data_at, lane, img);
	} else {
		constexpr int MID = (LO + HI) / 2;
		if (nb <= uint32_t(MID))
			fq_pack_dispatch<E, LO, MID>(nb, c, slots, data_at, lane, img);
		else
			fq_pack_dispatch<E, MID + 1, HI>(nb, c, slots, data_at, lane, img);
	}
}

// The encoder's four-block path with the image of the tile (LDS) as its
// destination: da, db are the halves of this lane's block descriptor, rel the
// block's place in the image.  The values of the blocks it cannot do are left
// in the slots as they are, and those blocks are reported as zero bits of the
// result.
template <typename E>
__device__ __forceinline__ uint32_t fq_encode(const typename E::C (&v)[VPQ], const uint4 da, const uint4 db,
					      uint32_t rel, int lane, typename E::C *slots, const uint4 *wt,
					      uint8_t *img)
{
	using C = typename E::C;
	static_assert(sizeof(C) == 4, "element types up to 32 bits");
	asm volatile("" : "+v"(lane));
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
	const bool ok = (enc == OROCH_BITPCK || enc == OROCH_BITFOR || enc == OROCH_BITPFR) && width_ok &&
			dataspace >= pck && size >= list_at && size - list_at <= uint32_t(TAIL_BYTES);
	const uint32_t fast = group_mask(__ballot(ok));
	const bool has_exc = ok && enc == OROCH_BITPFR;
	const bool zig = E::sgn && ok && enc == OROCH_BITPCK;
	const C org = ok && enc != OROCH_BITPCK ? origin : C(0);
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
	C *mine = slots + __umul24(uint32_t(g), SLOT_STRIDE);
#pragma unroll
	for (int k = 0; k < VPQ / 4; k++)
		*reinterpret_cast<uint4 *>(&mine[sub * VPQ + 4 * k]) = make_uint4(u[4 * k], u[4 * k + 1], u[4 * k + 2], u[4 * k + 3]);
	// the last packed word of a block reads past value 255: zeros there
	*reinterpret_cast<uint2 *>(&mine[BLOCK + 2 * sub]) = make_uint2(0, 0);
	if (fast == 0) {
		wave_sync();
		return 0;
	}
	// the header's eight or sixteen bytes and the lists' place start out as zeros
	uint8_t *bimg = img + rel;
	uint8_t *tail = bimg + list_at;
	const uint32_t tail_len = ok ? size - list_at : 0u;
	// (the lists' place is a multiple of 8 bytes long, not of 16)
	if (uint32_t(sub) * 16u < tail_len)
		*reinterpret_cast<uint2 *>(tail + 16 * sub) = make_uint2(0, 0);
	if (uint32_t(sub) * 16u + 8u < tail_len)
		*reinterpret_cast<uint2 *>(tail + 16 * sub + 8) = make_uint2(0, 0);
	uint32_t m = 0;
	if (__ballot(has_exc) != 0) {
		// u >= 2^nb exactly when u + (2^32 - 2^nb) carries
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
		fq_pack_dispatch<E, FIELDS_MIN_NB, E::W>(nb_u, make_class(cls), slots, rel + doff, lane, img);
		left &= ~cls;
	}

	// metadata, integer_codec.h:128-151, 177-182, 196-204: lane `sub` makes
	// byte `sub` of the (at most 11-byte) header and of the zeros behind it
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
		if (ok && uint32_t(sub) < doff)
			bimg[sub] = uint8_t(byte);
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
		const uint32_t ilen = ienc == OROCH_BITPCK ? (uint32_t(__umul24(nout + iw.x - 1u, iw.z)) >> 16) << 4
							    : lane_gather(isum, g * GROUP + GROUP - 1);
		uint32_t io = isum - ib, vo = ilen + vsum - vb, r = before;
		// (the lists start at a multiple of 8 bytes of the image)
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
	return fast;
}

// LDS per worker: a work area that holds the selection's tables during
// selection and the value slots during encoding, and the ring of finished tile
// images that wait for their offsets.
#ifndef FQ_RING
#define FQ_RING 6144
#endif
constexpr int FQ_RING_BYTES = FQ_RING;
#ifndef FQ_DEPTH
#define FQ_DEPTH 8
#endif
constexpr int FQ_QDEPTH = FQ_DEPTH; // rounds a workgroup may have under way (a power of two)
constexpr int FQ_QWORDS = 8;        // position, bytes, sizes of blocks 0..2

static_assert((FQ_QDEPTH & (FQ_QDEPTH - 1)) == 0, "rounds are taken modulo the depth");

// What the workers and the exchanger of a workgroup tell each other.  Round r
// sits at [r % depth].
struct fq_mail
{
	uint32_t bytes[FQ_QDEPTH][FQ_WORKERS];  // size of each worker's tile (the worker writes)
	uint32_t arrived[FQ_QDEPTH];            // workers that have written theirs << 24 | sum of sizes / 16
	uint64_t base[FQ_QDEPTH][FQ_WORKERS];   // offset of each worker's tile (the exchanger writes)
	uint32_t published;                     // rounds whose row entry has been stored
	uint32_t resolved;                      // rounds whose offset is in `base`
	uint32_t abort;                         // the launch has been given up
	uint32_t pad;
};

// Words in LDS that another wavefront of the workgroup writes or reads.  The
// LDS serves a wavefront's accesses in the order it issued them, so data
// written before a counter is there for whoever has seen the counter; all that
// is needed is that the compiler keeps the order too.  (Acquire and release
// forms would also wait for every global load in flight.)
__device__ __forceinline__ uint32_t fq_lds_load(const uint32_t *p)
{
	asm volatile("" ::: "memory");
	const uint32_t x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	asm volatile("" ::: "memory");
	return x;
}
__device__ __forceinline__ void fq_lds_store(uint32_t *p, uint32_t x)
{
	asm volatile("" ::: "memory");
	__hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	asm volatile("" ::: "memory");
}

template <typename E>
constexpr int fq_work_bytes()
{
	constexpr int sel = (QUAD * HQ_DWORDS + QUAD * NO_DWORDS) * 4 + QUAD * PW_ENTRIES * 16;
	constexpr int enc = int(sizeof(typename E::C)) * QUAD * SLOT_STRIDE;
	constexpr int m = sel > enc ? sel : enc;
	return (m + 15) / 16 * 16;
}

// A row of the table, four entries to a lane.
__device__ __forceinline__ uint64_t fq_row_load(const fq_state &st, uint32_t round, int lane)
{
	const uint64_t *p = reinterpret_cast<const uint64_t *>(st.rows + size_t(round) * FQ_MAX_GROUPS) + lane;
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The sum (in units of 16 bytes) of the first `count` entries of a row held four
// to a lane, if they are all there.
__device__ __forceinline__ bool fq_row_sum(uint64_t x, uint32_t count, int lane, uint32_t &sum)
{
	uint32_t part = 0;
	bool missing = false;
#pragma unroll
	for (int i = 0; i < 4; i++) {
		const uint32_t e = uint32_t(x >> (16 * i)) & 0xffffu;
		const bool wanted = uint32_t(4 * lane + i) < count;
		missing = missing || (wanted && !(e & FQ_ROW_VALID));
		part += wanted ? e & (FQ_ROW_VALID - 1u) : 0u;
	}
	if (__ballot(missing) != 0)
		return false;
	sum = wave_sum(part);
	return true;
}

// An exchanger whose wait has run out: the launch is given up (the abort word is
// set, the workgroup's workers are told), and the first exchanger to do so leaves
// behind, in the words after the abort word, where it stood: workgroup, round,
// which wait (1: its own workers' sizes, 2: the table), complete rows, rounds
// published and resolved, and the arrival words of the mail.
__device__ __forceinline__ void fq_give_up(const fq_state &st, fq_mail *mail, uint32_t group, uint32_t r, uint32_t wait,
					   uint32_t rows_done, int lane)
{
	if (lane == 0) {
		const uint32_t was = atomicCAS(st.abort_flag, 0u, 1u);
		if (was == 0) {
			uint32_t *note = st.abort_flag + 1;
			note[0] = group;
			note[1] = r;
			note[2] = wait;
			note[3] = rows_done;
			note[4] = mail->published;
			note[5] = mail->resolved;
			for (int i = 0; i < FQ_QDEPTH; i++)
				note[6 + i] = mail->arrived[i];
		}
	}
	fq_lds_store(&mail->abort, 1u);
}

// The exchanger's loop: for every round of the workgroup, once the workgroup's
// own entry is in the table, polls the table until the entries before it are
// there, and hands the offset to the workers.  Loads only.
__device__ __forceinline__ void fq_exchange(const fq_state &st, fq_mail *mail, uint32_t group, uint32_t groups,
					    uint32_t rounds, int lane)
{
	uint64_t base = 0;      // bytes of all tiles of the rounds before `rows_done`
	uint32_t rows_done = 0;
	uint32_t idle = 0;
	for (uint32_t r = 0; r < rounds; r++) {
		// the workgroup's own sizes first
		while (fq_lds_load(&mail->published) <= r) {
			if (++idle > FQ_IDLE_LIMIT || ((idle & 1023u) == 0 && __hip_atomic_load(st.abort_flag, __ATOMIC_RELAXED,
												  __HIP_MEMORY_SCOPE_AGENT))) {
				fq_give_up(st, mail, group, r, 1u, rows_done, lane);
				return;
			}
			__builtin_amdgcn_s_sleep(2);
		}
		for (;;) {
			FQ_COUNT(0, lane);
			const uint64_t t_loop = FQ_NOW();
			const bool need_prev = rows_done < r; // then rows_done == r - 1
			uint64_t prev = 0;
			if (need_prev)
				prev = fq_row_load(st, r - 1u, lane);
			uint64_t cur = 0;
			if (group != 0)
				cur = fq_row_load(st, r, lane);
			asm volatile("" : "+v"(prev), "+v"(cur));
			FQ_TIME(18, lane, t_loop);
			bool ok = true;
			if (need_prev) {
				uint32_t total = 0;
				ok = fq_row_sum(prev, groups, lane, total);
				if (ok) {
					base += uint64_t(total) << 4;
					rows_done = r;
				}
			}
			uint32_t before = 0;
			ok = ok && (group == 0 || fq_row_sum(cur, group, lane, before));
			if (ok) {
				// the workgroup's tiles of a round follow one another: every worker's
				// offset from the sizes the workers left (all final: the round is
				// published), so that no worker reads another's size later, when its
				// owner may be rounds ahead
				const uint32_t slot = r % uint32_t(FQ_QDEPTH);
				const uint32_t mine = lane < FQ_WORKERS ? mail->bytes[slot][lane] : 0u;
				const uint32_t upto = wave_scan<uint32_t>(mine, 0u, [](uint32_t a, uint32_t b) { return a + b; });
				if (lane < FQ_WORKERS)
					mail->base[slot][lane] = base + (uint64_t(before) << 4) + (upto - mine);
				fq_lds_store(&mail->resolved, r + 1u);
				FQ_COUNT(1, lane);
				idle = 0;
				break;
			}
			FQ_COUNT(5, lane);
			if (++idle > FQ_IDLE_LIMIT || ((idle & 63u) == 0 && __hip_atomic_load(st.abort_flag, __ATOMIC_RELAXED,
												__HIP_MEMORY_SCOPE_AGENT))) {
#ifdef FQ_STATS
				if (lane == 0 && group < 3)
					printf("exchanger %u gives up in round %u of %u\n", group, r, rounds);
#endif
				fq_give_up(st, mail, group, r, 2u, rows_done, lane);
				return;
			}
			__builtin_amdgcn_s_sleep(8);
		}
	}
}

template <typename E>
__global__ __launch_bounds__(FQ_THREADS) __attribute__((amdgpu_waves_per_eu(FQ_EU, FQ_EU))) void k_fused_quad(
	const typename E::S *__restrict__ src, size_t n, size_t nblocks, size_t ntiles,
	uint8_t *__restrict__ dst, uint64_t *__restrict__ offsets, fq_state st)
{
	using C = typename E::C;
	static_assert(FQ_RING_BYTES >= QUAD * (image_bytes<E>() - 16), "the ring holds any one tile");
	__shared__ __attribute__((aligned(16))) uint8_t work_area[FQ_WORKERS][fq_work_bytes<E>()];
	__shared__ __attribute__((aligned(16))) uint8_t rings[FQ_WORKERS][FQ_RING_BYTES];
	__shared__ uint32_t queues[FQ_WORKERS][FQ_QDEPTH][FQ_QWORDS];
	__shared__ fq_mail mail;
	__shared__ uint4 wt[65];
	const int wave = uniform(int(threadIdx.x / WAVE)), lane = lane_id();
	if (threadIdx.x < FQ_QDEPTH)
		mail.arrived[threadIdx.x] = 0;
	if (threadIdx.x == 0) {
		mail.published = 0;
		mail.resolved = 0;
		mail.abort = 0;
	}
	load_width_table(wt); // (ends with the workgroup's only barrier)
	const uint32_t group = blockIdx.x, groups = gridDim.x;
	const size_t per_round = size_t(groups) * FQ_WORKERS;
	// rounds of the workgroup = rounds of its first worker
	const size_t first0 = size_t(group) * FQ_WORKERS;
	const uint32_t wg_rounds = first0 < ntiles ? uint32_t((ntiles - first0 + per_round - 1) / per_round) : 0u;
#ifdef FQ_STATS
	__shared__ uint32_t dbgs[FQ_WORKERS + 1][32];
	if (lane < 32)
		dbgs[wave][lane] = 0;
	st.dbg = dbgs[wave];
	const uint64_t t_kernel = FQ_NOW();
	struct dbg_flush
	{
		uint32_t *d;
		int lane, wave;
		uint64_t t0;
		__device__ ~dbg_flush()
		{
			if (lane == 0)
				d[wave < FQ_WORKERS ? 11 : 15] += uint32_t((__builtin_amdgcn_s_memtime() - t0) >> 4);
			asm volatile("" ::: "memory");
			if (lane < 32)
				atomicAdd(&fq_stats[lane], (unsigned long long) d[lane]);
		}
	} dbg_at_exit{dbgs[wave], lane, wave, t_kernel};
#endif
	if (wave == FQ_WORKERS) {
		// (little to do and others wait for it: ahead of the workers when it has something)
		__builtin_amdgcn_s_setprio(3);
		fq_exchange(st, &mail, group, groups, wg_rounds, lane);
		return;
	}
	const lane_consts lc = make_lane_consts(wt, lane);
	const quad_lane_consts qlc = make_quad_lane_consts(wt, lane);
	// selection's view of the work area
	uint32_t *hq = reinterpret_cast<uint32_t *>(work_area[wave]);
	uint32_t *no = hq + QUAD * HQ_DWORDS;
	uint4 *pw = reinterpret_cast<uint4 *>(no + QUAD * NO_DWORDS);
	// the encoder's view
	C *slots = reinterpret_cast<C *>(work_area[wave]);
	uint8_t *ring = rings[wave];
	uint32_t(*queue)[FQ_QWORDS] = queues[wave];

	const int g = int(uint32_t(lane) >> 4), sub = int(uint32_t(lane) & 15u);
	const size_t first = first0 + size_t(wave);
	const uint32_t rounds = first < ntiles ? uint32_t((ntiles - first + per_round - 1) / per_round) : 0u;

	// the ring: tiles in the order they were finished, each in one piece
	uint32_t flushed = 0, wpos = 0; // rounds copied out; where the next image goes
	// copies the oldest tile (round `flushed`) out at its offset and records where its blocks start
	auto flush = [&]() {
#ifdef FQ_WAIT_LOADS
		// the values asked for ahead must have arrived before anything is stored:
		// behind the stores they would wait for the stores to be acknowledged
		__builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0)
#endif
		const uint32_t slot = flushed % uint32_t(FQ_QDEPTH);
		const size_t t = first + size_t(flushed) * per_round;
		const uint64_t base = uniform(mail.base[slot][wave]);
		const uint32_t pos = uniform(queue[slot][0]), total = uniform(queue[slot][1]);
		const uint32_t s0 = uniform(queue[slot][2]), s1 = uniform(queue[slot][3]), s2 = uniform(queue[slot][4]);
		for (uint32_t o = uint32_t(lane) * 16u; o < total; o += WAVE * 16u)
			*reinterpret_cast<uint4 *>(dst + base + o) = *reinterpret_cast<const uint4 *>(ring + pos + o);
		const size_t b0 = t * QUAD;
		const uint32_t rel = lane == 0 ? 0u : lane == 1 ? s0 : lane == 2 ? s0 + s1 : s0 + s1 + s2;
		if (lane < QUAD && b0 + size_t(lane) < nblocks)
			offsets[b0 + size_t(lane)] = base + rel;
		if (t == ntiles - 1 && lane == 0)
			offsets[nblocks] = base + total;
		flushed++;
	};
	// copies out what has an offset by now
	auto flush_ready = [&](uint32_t upto) {
		const uint32_t have = fq_lds_load(&mail.resolved);
		while (flushed != upto && flushed < have)
			flush();
	};
	// waits until the oldest tile has its offset and copies it out; false if the launch was given up
	auto flush_oldest = [&]() -> bool {
		FQ_COUNT(4, lane);
		const uint64_t t_block = FQ_NOW();
		while (fq_lds_load(&mail.resolved) <= flushed) {
			if (fq_lds_load(&mail.abort))
				return false;
			__builtin_amdgcn_s_sleep(2);
		}
		FQ_TIME(10, lane, t_block);
		flush();
		return true;
	};
	// a place for `total` bytes in one piece when rounds [flushed, r) are in the ring, or ~0
	auto place = [&](uint32_t r, uint32_t total) -> uint32_t {
		if (r == flushed)
			return 0;
		const uint32_t rpos = uniform(queue[flushed % uint32_t(FQ_QDEPTH)][0]);
		if (wpos > rpos) {
			if (wpos + total <= uint32_t(FQ_RING_BYTES))
				return wpos;
			return total <= rpos ? 0u : ~0u;
		}
		return wpos < rpos && wpos + total <= rpos ? wpos : ~0u;
	};

	// The values of the next round are loaded as soon as this round's selection is done.
	auto is_whole = [&](size_t t) -> bool {
		return t < ntiles && t * QUAD + QUAD <= nblocks && (t * QUAD + QUAD) * BLOCK <= n;
	};
	// (both start out defined: an array that is only written under a condition
	// is kept in scratch memory, not in registers)
	C v[VPQ] = {}, vn[VPQ] = {};
	if (is_whole(first))
		load16<E>(src + (first * QUAD + size_t(g)) * BLOCK + size_t(sub) * VPQ, vn);
	for (uint32_t r = 0; r < rounds; r++) {
		const size_t tile = first + size_t(r) * per_round;
		const size_t blk0 = tile * QUAD;
		const bool whole = is_whole(tile);
		uint4 da = make_uint4(0, 0, 0, 0), db = da;
		uint32_t size = 0; // of block g (whole quads) or of block `lane` (others)
		if (whole) {
			quad_selection sel;
			const uint64_t t_v = FQ_NOW();
#pragma unroll
			for (int i = 0; i < VPQ; i++)
				v[i] = vn[i];
#ifdef FQ_STATS
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
			FQ_TIME(12, lane, t_v);
			const uint64_t t_sel = FQ_NOW();
			select_quad<E>(v, lane, hq, no, pw, wt, qlc, sel);
			FQ_TIME(13, lane, t_sel);
			wave_sync();
			desc_words(sel, da, db);
			size = da.w;
		} else {
			for (int b = 0; b < QUAD && blk0 + size_t(b) < nblocks; b++) {
				const size_t blk = blk0 + size_t(b);
				const int nv = block_count(n, blk);
				C w[VPL];
				load_block<E>(src, blk * BLOCK, nv, lane, w);
				selection<C> s1;
				select_any<E>(w, nv, lane, hq, wt, lc, s1);
				size = lane == b * GROUP ? stream_size(s1) : size;
				wave_sync();
			}
		}
		const uint32_t s0 = read_lane(size, 0), s1 = read_lane(size, GROUP), s2 = read_lane(size, 2 * GROUP),
			       s3 = read_lane(size, 3 * GROUP);
		const uint32_t total = s0 + s1 + s2 + s3;
		const uint32_t rel = g == 0 ? 0u : g == 1 ? s0 : g == 2 ? s0 + s1 : s0 + s1 + s2;
		// next round's values
		const uint64_t t_mid = FQ_NOW();
		if (is_whole(tile + per_round))
			load16<E>(src + ((tile + per_round) * QUAD + size_t(g)) * BLOCK + size_t(sub) * VPQ, vn);
		// this tile's size (the mail has a place per round under way); the worker
		// that is last with its size stores the workgroup's entry of the table
		flush_ready(r);
		while (r - flushed >= uint32_t(FQ_QDEPTH)) {
			if (!flush_oldest())
				return;
		}
		{
			const uint32_t slot = r % uint32_t(FQ_QDEPTH);
			// workers that have this round: all twelve but at the array's end
			const size_t tile0 = first0 + size_t(r) * per_round;
			const uint32_t nw = ntiles - tile0 < size_t(FQ_WORKERS) ? uint32_t(ntiles - tile0) : uint32_t(FQ_WORKERS);
			uint32_t old = 0;
			if (lane == 0) {
				mail.bytes[slot][wave] = total;
				asm volatile("" ::: "memory");
				old = __hip_atomic_fetch_add(&mail.arrived[slot], (1u << 24) | (total >> 4), __ATOMIC_RELAXED,
							     __HIP_MEMORY_SCOPE_WORKGROUP);
			}
			old = uniform(old);
			if ((old >> 24) == nw - 1u) {
				const uint32_t sum16 = (old & 0xffffffu) + (total >> 4);
				if (lane == 0) {
					mail.arrived[slot] = 0;
					__hip_atomic_store(&st.rows[size_t(r) * FQ_MAX_GROUPS + group], uint16_t(FQ_ROW_VALID | sum16),
							   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				// (a maximum, not a store: in the array's last round, which not
				// every worker has, the round's last worker can get here ahead
				// of the last worker of the round before)
				asm volatile("" ::: "memory");
				if (lane == 0)
					__hip_atomic_fetch_max(&mail.published, r + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				asm volatile("" ::: "memory");
			}
		}
		FQ_TIME(2, lane, t_mid);
		// room in the ring
		uint32_t pos;
		while ((pos = place(r, total)) == ~0u) {
			if (!flush_oldest())
				return;
		}
		uint8_t *img = ring + pos;
		FQ_TIME(3, lane, t_mid);
		const uint64_t t_enc = FQ_NOW();
		uint32_t done = 0;
		if (whole)
			done = fq_encode<E>(v, da, db, rel, lane, slots, wt, img);
		if (done != 15u) {
			for (int b = 0; b < QUAD && blk0 + size_t(b) < nblocks; b++) {
				if ((done >> b) & 1u)
					continue;
				const size_t blk = blk0 + size_t(b);
				const int nv = block_count(n, blk);
				C w[VPL];
				selection<C> s1;
				const int l = b * GROUP;
				if (whole) {
					// the raw values wait in the slots
					const uint4 q = *reinterpret_cast<const uint4 *>(&slots[b * SLOT_STRIDE + lane * VPL]);
					w[0] = q.x, w[1] = q.y, w[2] = q.z, w[3] = q.w;
					s1.origin = C(read_lane(da.x, l));
					s1.dataspace = read_lane(da.z, l);
					const uint32_t w4 = read_lane(db.x, l), w5 = read_lane(db.y, l), w6 = read_lane(db.z, l);
					s1.metaspace = w4 & 0xffffu;
					s1.noutliers = w4 >> 16;
					s1.encoding = (w5 >> 16) & 0xffu;
					s1.nbits = w5 >> 24;
					s1.ienc = w6 & 0xffu;
					s1.inbits = (w6 >> 8) & 0xffu;
					s1.venc = (w6 >> 16) & 0xffu;
					s1.vnbits = w6 >> 24;
				} else {
					load_block<E>(src, blk * BLOCK, nv, lane, w);
					select_any<E>(w, nv, lane, hq, wt, lc, s1);
					wave_sync();
				}
				uint32_t *bimg = reinterpret_cast<uint32_t *>(img + read_lane(rel, l));
				if (nv == BLOCK)
					(void) encode_block<E, true>(w, nv, lane, s1, wt, bimg);
				else
					(void) encode_block<E, false>(w, nv, lane, s1, wt, bimg);
				wave_sync();
			}
		}
		FQ_TIME(14, lane, t_enc);
		// the tile joins the ring
		{
			const uint32_t slot = r % uint32_t(FQ_QDEPTH);
			if (lane == 0) {
				queue[slot][0] = pos;
				queue[slot][1] = total;
				queue[slot][2] = s0;
				queue[slot][3] = s1;
				queue[slot][4] = s2;
			}
			wpos = pos + total;
			wave_sync();
		}
	}
	while (flushed != rounds) {
		if (!flush_oldest())
			return;
	}
}

// Workgroups of k_fused_quad the launch may have: all must be resident (a tile
// may wait for any lower-numbered one), and a row of the table holds 256.
template <typename E>
static int fq_resident_workgroups()
{
	static int cached = 0;
	if (cached)
		return cached;
	int dev = 0, cus = 0, per_cu = 0;
	if (hipGetDevice(&dev) != hipSuccess ||
	    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
	    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fused_quad<E>, FQ_THREADS, 0) != hipSuccess)
		return -1;
	if (per_cu < 1 || cus < 1)
		return -1;
	// one per CU even where a second would fit: thirteen wavefronts leave no
	// even share of the SIMDs for two workgroups
	cached = cus < FQ_MAX_GROUPS ? cus : FQ_MAX_GROUPS;
	return cached;
}

} // namespace oroch_hip
