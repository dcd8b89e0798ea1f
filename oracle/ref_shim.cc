// ref_shim.cc -- C entry points over the REAL reference headers.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no codec logic of its own: it
// instantiates the reference's templates (included from where they lie,
// -I/root/reference, see oracle/Makefile) for the eight element types and
// exposes them with the same flat signatures as oracle/oroch_oracle.h, so the
// C restatement, the golden fixtures and the CPU baseline can all be driven
// from Python/ctypes.  It is built only into oracle/_ref/ and never linked
// into anything that ships.
#include <oroch/bitfor.h>
#include <oroch/bitpck.h>
#include <oroch/bitpfr.h>
#include <oroch/integer_array.h>
#include <oroch/integer_codec.h>
#include <oroch/integer_group.h>
#include <oroch/integer_stats.h>
#include <oroch/varint.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <sstream>
#include <thread>
#include <vector>

#include "oroch_oracle.h" // orc_meta record layout only

namespace {

template <typename T>
struct ref
{
	using U = typename oroch::integer_traits<T>::unsigned_t;
	using codec = oroch::integer_codec<T>;
	using metadata = typename codec::metadata;

	static void to_flat(const metadata &m, orc_meta *o)
	{
		std::memset(o, 0, sizeof *o);
		o->encoding = m.value_desc.encoding;
		o->nbits = uint32_t(m.value_desc.nbits);
		o->origin = uint64_t(U(m.value_desc.origin));
		o->dataspace = m.dataspace();
		o->metaspace = m.metaspace();
		o->noutliers = m.noutliers;
		o->index_encoding = m.outlier_index_desc.encoding;
		o->index_nbits = uint32_t(m.outlier_index_desc.nbits);
		o->value_encoding = m.outlier_value_desc.encoding;
		o->value_nbits = uint32_t(m.outlier_value_desc.nbits);
	}

	static void from_flat(const orc_meta *o, metadata &m)
	{
		m.clear();
		m.value_desc.encoding = oroch::encoding_t(o->encoding);
		m.value_desc.nbits = o->nbits;
		m.value_desc.origin = T(U(o->origin));
		m.value_desc.dataspace = o->dataspace;
		m.value_desc.metaspace = o->metaspace - 1;
		m.noutliers = o->noutliers;
		m.outlier_index_desc.encoding = oroch::encoding_t(o->index_encoding);
		m.outlier_index_desc.nbits = o->index_nbits;
		m.outlier_value_desc.encoding = oroch::encoding_t(o->value_encoding);
		m.outlier_value_desc.nbits = o->value_nbits;
	}

	static size_t bitpck_encode(const T *src, size_t n, unsigned nbits, uint8_t *dst)
	{
		oroch::dst_bytes_t d = dst;
		oroch::bitpck_codec<T>::encode(d, src, src + n, nbits);
		return d - dst;
	}
	static size_t bitpck_decode(const uint8_t *src, size_t n, unsigned nbits, T *dst)
	{
		oroch::src_bytes_t s = src;
		oroch::bitpck_codec<T>::decode(dst, dst + n, s, nbits);
		return s - src;
	}
	static size_t bitfor_encode(const T *src, size_t n, T origin, unsigned nbits, uint8_t *dst)
	{
		oroch::dst_bytes_t d = dst;
		typename oroch::bitfor_codec<T>::parameters p(origin, nbits);
		oroch::bitfor_codec<T>::encode(d, src, src + n, p);
		return d - dst;
	}
	static size_t bitfor_decode(const uint8_t *src, size_t n, T origin, unsigned nbits, T *dst)
	{
		oroch::src_bytes_t s = src;
		typename oroch::bitfor_codec<T>::parameters p(origin, nbits);
		oroch::bitfor_codec<T>::decode(dst, dst + n, s, p);
		return s - src;
	}
	static size_t bitpfr_encode(const T *src, size_t n, T origin, unsigned nbits, uint8_t *dst,
				    uint64_t *indices, uint64_t *values, size_t *nexc)
	{
		typename oroch::bitpfr_codec<T>::exceptions x;
		typename oroch::bitpfr_codec<T>::parameters p(origin, nbits, x);
		oroch::dst_bytes_t d = dst;
		oroch::bitpfr_codec<T>::encode(d, src, src + n, p);
		for (size_t i = 0; i < x.indices.size(); i++) {
			indices[i] = x.indices[i];
			values[i] = x.values[i];
		}
		*nexc = x.indices.size();
		return d - dst;
	}
	static size_t bitpfr_decode(const uint8_t *src, size_t n, T origin, unsigned nbits,
				    const uint64_t *indices, const uint64_t *values, size_t nexc,
				    T *dst)
	{
		typename oroch::bitpfr_codec<T>::exceptions x;
		for (size_t i = 0; i < nexc; i++) {
			x.indices.push_back(indices[i]);
			x.values.push_back(U(values[i]));
		}
		typename oroch::bitpfr_codec<T>::parameters p(origin, nbits, x);
		oroch::src_bytes_t s = src;
		oroch::bitpfr_codec<T>::decode(dst, dst + n, s, p);
		return s - src;
	}
	static size_t varint_encode(const T *src, size_t n, uint8_t *dst)
	{
		oroch::dst_bytes_t d = dst;
		oroch::varint_codec<T>::encode(d, src, src + n);
		return d - dst;
	}
	static size_t varint_decode(const uint8_t *src, size_t n, T *dst)
	{
		oroch::src_bytes_t s = src;
		oroch::varint_codec<T>::decode(dst, dst + n, s);
		return s - src;
	}
	static void stats(const T *src, size_t n, uint64_t *mn, uint64_t *mx, uint64_t *hist)
	{
		oroch::integer_stats<T> st(src, src + n);
		*mn = uint64_t(U(st.min()));
		*mx = uint64_t(U(st.max()));
		if (hist) {
			std::memset(hist, 0, 65 * sizeof *hist);
			st.build_histogram(src, src + n);
			for (size_t i = 0; i <= oroch::integer_stats<T>::nbits; i++)
				hist[i] = st.histogram(i);
		}
	}
	static void select(const T *src, size_t n, orc_meta *o)
	{
		metadata m;
		codec::select(m, src, src + n);
		to_flat(m, o);
	}
	static size_t meta_encode(const orc_meta *o, uint8_t *dst)
	{
		metadata m;
		from_flat(o, m);
		oroch::dst_bytes_t d = dst;
		m.encode(d);
		return d - dst;
	}
	static size_t meta_decode(const uint8_t *src, orc_meta *o)
	{
		metadata m;
		oroch::src_bytes_t s = src;
		m.decode(s);
		to_flat(m, o);
		o->dataspace = 0;
		o->metaspace = 0;
		return s - src;
	}
	static size_t codec_encode(const T *src, size_t n, const orc_meta *o, uint8_t *dst)
	{
		metadata m;
		from_flat(o, m);
		oroch::dst_bytes_t d = dst;
		codec::encode(d, src, src + n, m);
		return d - dst;
	}
	static size_t codec_decode(const uint8_t *src, size_t n, const orc_meta *o, T *dst)
	{
		metadata m;
		from_flat(o, m);
		oroch::src_bytes_t s = src;
		codec::decode(dst, dst + n, s, m);
		return s - src;
	}

	// integer_group keeps its buffer protected; a subclass reads it out.
	struct group : public oroch::integer_group<T>
	{
		const uint8_t *bytes() const
		{
			return this->data_.get();
		}
		void adopt(const uint8_t *src, size_t len)
		{
			this->data_.reset(new oroch::byte_t[len]);
			std::memcpy(this->data_.get(), src, len);
		}
	};

	// The real integer_group::encode, copied out with the pad zeroed.
	static size_t group_encode(const T *src, size_t n, uint8_t *dst)
	{
		group g;
		g.encode(src, src + n);
		metadata m;
		codec::select(m, src, src + n);
		size_t meta = m.metaspace();
		size_t off = (meta + 7) & ~size_t(7);
		size_t len = off + m.dataspace();
		std::memcpy(dst, g.bytes(), len);
		std::memset(dst + meta, 0, off - meta);
		return len;
	}
	static size_t group_decode(const uint8_t *src, size_t len, size_t n, T *dst)
	{
		group g;
		g.adopt(src, len);
		g.decode(dst, dst + n);
		return len;
	}

	static size_t blocks_encode(const T *src, size_t n, size_t block, size_t align, uint8_t *dst,
				    uint64_t *offsets)
	{
		size_t nblocks = (n + block - 1) / block;
		size_t pos = 0;
		for (size_t b = 0; b < nblocks; b++) {
			size_t cnt = std::min(block, n - b * block);
			offsets[b] = pos;
			size_t len = group_encode(src + b * block, cnt, dst + pos);
			size_t end = (pos + len + align - 1) / align * align;
			std::memset(dst + pos + len, 0, end - (pos + len));
			pos = end;
		}
		offsets[nblocks] = pos;
		return pos;
	}

	//
	// CPU baseline: the loop shape of tests/bench/group-bench.cc applied to
	// blocks of `block` values, phases timed separately.  Blocks are split
	// statically over `threads` std::threads (the reference has no threading
	// of its own).  Returns seconds per phase in out[0..2] = select, encode
	// (metadata + data, given the selection), decode.
	//
	// Wall time of the whole-sequence codecs with the caller's parameters on
	// one thread (the loop shape of tests/bench/group-bench.cc: buffers made
	// once, then the calls timed): kind 0 bitpck, 1 bitfor, 2 bitpfr.
	// out[0] = seconds to encode, out[1] = seconds to decode, out[2] = 1 when the
	// values came back.
	static void bench_whole(const T *src, size_t n, int kind, T origin, unsigned nbits, double *out)
	{
		std::vector<uint8_t> bytes(oroch::bitpck_codec<T>::space(n, nbits) + 16);
		std::vector<T> back(n);
		typename oroch::bitpfr_codec<T>::exceptions x;
		auto now = [] { return std::chrono::steady_clock::now(); };
		auto t0 = now();
		oroch::dst_bytes_t d = bytes.data();
		if (kind == 0) {
			oroch::bitpck_codec<T>::encode(d, src, src + n, nbits);
		} else if (kind == 1) {
			typename oroch::bitfor_codec<T>::parameters p(origin, nbits);
			oroch::bitfor_codec<T>::encode(d, src, src + n, p);
		} else {
			typename oroch::bitpfr_codec<T>::parameters p(origin, nbits, x);
			oroch::bitpfr_codec<T>::encode(d, src, src + n, p);
		}
		auto t1 = now();
		oroch::src_bytes_t s = bytes.data();
		if (kind == 0) {
			oroch::bitpck_codec<T>::decode(back.begin(), back.end(), s, nbits);
		} else if (kind == 1) {
			typename oroch::bitfor_codec<T>::parameters p(origin, nbits);
			oroch::bitfor_codec<T>::decode(back.begin(), back.end(), s, p);
		} else {
			typename oroch::bitpfr_codec<T>::parameters p(origin, nbits, x);
			oroch::bitpfr_codec<T>::decode(back.begin(), back.end(), s, p);
		}
		auto t2 = now();
		out[0] = std::chrono::duration<double>(t1 - t0).count();
		out[1] = std::chrono::duration<double>(t2 - t1).count();
		out[2] = std::equal(back.begin(), back.end(), src) ? 1.0 : 0.0;
	}

	static void bench_blocks(const T *src, size_t n, size_t block, int threads, double *out)
	{
		size_t nblocks = n / block;
		std::vector<metadata> metas(nblocks);
		std::vector<size_t> offs(nblocks + 1);
		auto run = [&](auto fn) {
			auto t0 = std::chrono::steady_clock::now();
			std::vector<std::thread> pool;
			for (int t = 0; t < threads; t++) {
				size_t lo = nblocks * t / threads, hi = nblocks * (t + 1) / threads;
				pool.emplace_back([=, &fn] {
					for (size_t b = lo; b < hi; b++)
						fn(b);
				});
			}
			for (auto &th : pool)
				th.join();
			return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0)
				.count();
		};
		out[0] = run([&](size_t b) {
			metadata m;
			codec::select(m, src + b * block, src + (b + 1) * block);
			metas[b] = m;
		});
		size_t pos = 0;
		for (size_t b = 0; b < nblocks; b++) {
			offs[b] = pos;
			pos += ((metas[b].metaspace() + 7) & ~size_t(7)) + metas[b].dataspace();
			pos = (pos + 15) & ~size_t(15);
		}
		offs[nblocks] = pos;
		std::vector<uint8_t> bytes(pos + 16);
		out[1] = run([&](size_t b) {
			oroch::dst_bytes_t d = bytes.data() + offs[b];
			metas[b].encode(d);
			d = bytes.data() + offs[b] + ((metas[b].metaspace() + 7) & ~size_t(7));
			codec::encode(d, src + b * block, src + (b + 1) * block, metas[b]);
		});
		std::vector<T> back(nblocks * block);
		out[2] = run([&](size_t b) {
			metadata m;
			oroch::src_bytes_t s = bytes.data() + offs[b];
			oroch::src_bytes_t s0 = s;
			m.decode(s);
			size_t off = (size_t(s - s0) + 7) & ~size_t(7);
			s = s0 + off;
			codec::decode(back.data() + b * block, back.data() + (b + 1) * block, s, m);
		});
		out[3] = double(pos);
		out[4] = std::memcmp(back.data(), src, nblocks * block * sizeof(T)) == 0 ? 1.0 : 0.0;
	}
};

} // namespace

// The reference's integer_array (integer_array.h:150-259) through a script:
// values[0..n) are inserted at the end one by one, then nops insertions at
// the given positions follow, then every needle is looked up with find().
// found[j] receives the position (or not_found); info receives the text of
// group_info() (one line of metadata per full group) and its length is returned.
// taken[k] tells whether insertion k went through: the reference's range check
// (integer_array.h:221) throws std::out_of_range when the position within its
// group exceeds the length of the tail, also for positions inside the array.
template <typename T>
static size_t array_script(const T *values, size_t n, const uint64_t *ins_pos, const T *ins_val,
			   size_t nops, uint8_t *taken, const T *needles, size_t m, uint64_t *found,
			   char *info, size_t info_capacity)
{
	oroch::integer_array<T> array;
	for (size_t i = 0; i < n; i++)
		array.insert(i, values[i]);
	for (size_t k = 0; k < nops; k++) {
		taken[k] = 1;
		try {
			array.insert(size_t(ins_pos[k]), ins_val[k]);
		} catch (const std::out_of_range &) {
			taken[k] = 0;
		}
	}
	for (size_t j = 0; j < m; j++)
		found[j] = uint64_t(array.find(needles[j]));
	std::ostringstream os;
	array.group_info(os);
	const std::string text = os.str();
	const size_t len = text.size() < info_capacity ? text.size() : info_capacity;
	std::memcpy(info, text.data(), len);
	return len;
}

#define DISPATCH(dtype, expr)                                                                      \
	switch (dtype) {                                                                           \
	case 0: { using T = int8_t; return expr; }                                                 \
	case 1: { using T = uint8_t; return expr; }                                                \
	case 2: { using T = int16_t; return expr; }                                                \
	case 3: { using T = uint16_t; return expr; }                                               \
	case 4: { using T = int32_t; return expr; }                                                \
	case 5: { using T = uint32_t; return expr; }                                               \
	case 6: { using T = int64_t; return expr; }                                                \
	default: { using T = uint64_t; return expr; }                                              \
	}

extern "C" {

size_t ref_bitpck_space(size_t n, size_t nbits)
{
	return n ? oroch::bitpck_codec<uint64_t>::space(n, nbits) : 0;
}
size_t ref_bitpck_encode(int dtype, const void *src, size_t n, unsigned nbits, uint8_t *dst)
{
	DISPATCH(dtype, ref<T>::bitpck_encode((const T *) src, n, nbits, dst));
}
size_t ref_bitpck_decode(int dtype, const uint8_t *src, size_t n, unsigned nbits, void *dst)
{
	DISPATCH(dtype, ref<T>::bitpck_decode(src, n, nbits, (T *) dst));
}
size_t ref_bitfor_encode(int dtype, const void *src, size_t n, uint64_t origin, unsigned nbits,
			 uint8_t *dst)
{
	DISPATCH(dtype, ref<T>::bitfor_encode((const T *) src, n, T(origin), nbits, dst));
}
size_t ref_bitfor_decode(int dtype, const uint8_t *src, size_t n, uint64_t origin, unsigned nbits,
			 void *dst)
{
	DISPATCH(dtype, ref<T>::bitfor_decode(src, n, T(origin), nbits, (T *) dst));
}
size_t ref_bitpfr_encode(int dtype, const void *src, size_t n, uint64_t origin, unsigned nbits,
			 uint8_t *dst, uint64_t *indices, uint64_t *values, size_t *nexc)
{
	DISPATCH(dtype, ref<T>::bitpfr_encode((const T *) src, n, T(origin), nbits, dst, indices,
					      values, nexc));
}
size_t ref_bitpfr_decode(int dtype, const uint8_t *src, size_t n, uint64_t origin, unsigned nbits,
			 const uint64_t *indices, const uint64_t *values, size_t nexc, void *dst)
{
	DISPATCH(dtype, ref<T>::bitpfr_decode(src, n, T(origin), nbits, indices, values, nexc,
					      (T *) dst));
}
size_t ref_varint_encode(int dtype, const void *src, size_t n, uint8_t *dst)
{
	DISPATCH(dtype, ref<T>::varint_encode((const T *) src, n, dst));
}
size_t ref_varint_decode(int dtype, const uint8_t *src, size_t n, void *dst)
{
	DISPATCH(dtype, ref<T>::varint_decode(src, n, (T *) dst));
}
void ref_stats(int dtype, const void *src, size_t n, uint64_t *mn, uint64_t *mx, uint64_t *hist)
{
	DISPATCH(dtype, ref<T>::stats((const T *) src, n, mn, mx, hist));
}
void ref_select(int dtype, const void *src, size_t n, orc_meta *meta)
{
	DISPATCH(dtype, ref<T>::select((const T *) src, n, meta));
}
size_t ref_meta_encode(int dtype, const orc_meta *meta, uint8_t *dst)
{
	DISPATCH(dtype, ref<T>::meta_encode(meta, dst));
}
size_t ref_meta_decode(int dtype, const uint8_t *src, orc_meta *meta)
{
	DISPATCH(dtype, ref<T>::meta_decode(src, meta));
}
size_t ref_codec_encode(int dtype, const void *src, size_t n, const orc_meta *meta, uint8_t *dst)
{
	DISPATCH(dtype, ref<T>::codec_encode((const T *) src, n, meta, dst));
}
size_t ref_codec_decode(int dtype, const uint8_t *src, size_t n, const orc_meta *meta, void *dst)
{
	DISPATCH(dtype, ref<T>::codec_decode(src, n, meta, (T *) dst));
}
size_t ref_group_encode(int dtype, const void *src, size_t n, uint8_t *dst)
{
	DISPATCH(dtype, ref<T>::group_encode((const T *) src, n, dst));
}
size_t ref_group_decode(int dtype, const uint8_t *src, size_t len, size_t n, void *dst)
{
	DISPATCH(dtype, ref<T>::group_decode(src, len, n, (T *) dst));
}
size_t ref_blocks_encode(int dtype, const void *src, size_t n, size_t block, size_t align,
			 uint8_t *dst, uint64_t *offsets)
{
	DISPATCH(dtype, ref<T>::blocks_encode((const T *) src, n, block, align, dst, offsets));
}
size_t ref_array_script(int dtype, const void *values, size_t n, const uint64_t *ins_pos,
			const void *ins_val, size_t nops, uint8_t *taken, const void *needles,
			size_t m, uint64_t *found, char *info, size_t info_capacity)
{
	DISPATCH(dtype, array_script<T>((const T *) values, n, ins_pos, (const T *) ins_val, nops, taken,
					(const T *) needles, m, found, info, info_capacity));
}
void ref_bench_whole(int dtype, const void *src, size_t n, int kind, uint64_t origin, unsigned nbits, double *out)
{
	DISPATCH(dtype, ref<T>::bench_whole((const T *) src, n, kind, T(origin), nbits, out));
}
void ref_bench_blocks(int dtype, const void *src, size_t n, size_t block, int threads, double *out)
{
	DISPATCH(dtype, ref<T>::bench_blocks((const T *) src, n, block, threads, out));
}

} // extern "C"
