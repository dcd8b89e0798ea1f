// host_templates_test.cc -- exercises the C++ host templates of include/oroch/
// on the GPU and checks every byte against the CPU oracle.  The scenarios
// follow the reference's unit tests (tests/unit/bitpck.cc, bitfor.cc,
// bitpfr.cc, integer_codec.cc, integer_group.cc) so that a reader can put the
// two side by side; the oracle comparisons go beyond them.
#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <sstream>
#include <deque>
#include <vector>

#include <oroch/integer_array.h>
#include <oroch/integer_container.h>
#include <oroch/integer_group.h>

#include "oroch_oracle.h"

static int g_checks = 0, g_failed = 0;

#define REQUIRE(cond)                                                                        \
	do {                                                                                 \
		g_checks++;                                                                  \
		if (!(cond)) {                                                               \
			g_failed++;                                                          \
			std::fprintf(stderr, "%s:%d: REQUIRE(%s) failed\n", __FILE__, __LINE__, #cond); \
			if (g_failed > 20)                                                   \
				std::exit(1);                                                \
		}                                                                            \
	} while (0)

static uint64_t g_rng = 0x9e3779b97f4a7c15ull;
static uint64_t rnd()
{
	uint64_t z = (g_rng += 0x9e3779b97f4a7c15ull);
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
	return z ^ (z >> 31);
}

template <typename T>
static std::vector<T> make_block(int shape, size_t n)
{
	using U = typename std::make_unsigned<T>::type;
	constexpr int W = sizeof(T) * 8;
	std::vector<T> v(n);
	const U base = U(rnd());
	for (size_t i = 0; i < n; i++) {
		U u = 0;
		switch (shape) {
		case 0: u = base; break;                                   // constant
		case 1: u = U(rnd()); break;                               // full range
		case 2: u = U(rnd() & 0x3f); break;                        // small unsigned
		case 3: u = U(base + (rnd() & 0x1f)); break;               // narrow range far from zero
		case 4:                                                    // body plus outliers
			u = U(base + (rnd() % 100 < 4 ? U(rnd()) >> (rnd() % (W / 2)) : U(rnd() & 0x7)));
			break;
		case 5: u = U(T(int64_t(rnd() % 200) - 100)); break;       // small signed
		case 6: u = U(i * 3); break;                               // ramp
		default: u = U(rnd() % 3 == 0 ? rnd() : 0); break;         // sparse
		}
		v[i] = T(u);
	}
	return v;
}

template <typename T>
static void test_stats_select_group()
{
	const int dt = oroch::hip::dtype_of<T>();
	constexpr int W = sizeof(T) * 8;
	// up to a block: the block kernels; beyond: one group over the whole-sequence kernels
	const size_t sizes[] = {1, 2, 4, 5, 8, 31, 100, 255, 256, 257, 1000, 4096};
	for (int shape = 0; shape < 8; shape++) {
		for (size_t n : sizes) {
			const std::vector<T> v = make_block<T>(shape, n);

			// integer_stats against the oracle.
			uint64_t mn, mx, hist[65];
			orc_stats(dt, v.data(), n, &mn, &mx, hist);
			oroch::integer_stats<T> st(v.begin(), v.end());
			st.build_histogram(v.begin(), v.end());
			REQUIRE(st.nvalues() == n);
			REQUIRE(oroch::hip::raw64<T>(st.min()) == mn);
			REQUIRE(oroch::hip::raw64<T>(st.max()) == mx);
			for (int b = 0; b <= W; b++)
				REQUIRE(st.histogram(b) == hist[b]);

			// integer_codec::select against the oracle.
			orc_meta om;
			orc_select(dt, v.data(), n, &om);
			typename oroch::integer_codec<T>::metadata meta;
			oroch::integer_codec<T>::select(meta, v.begin(), v.end());
			REQUIRE(meta.value_desc.encoding == om.encoding);
			REQUIRE(meta.dataspace() == om.dataspace);
			REQUIRE(meta.metaspace() == om.metaspace);
			if (om.encoding != oroch::normal && om.encoding != oroch::varint
			    && om.encoding != oroch::bitpck)
				REQUIRE(oroch::hip::raw64<T>(meta.value_desc.origin) == om.origin);
			if (om.encoding >= oroch::bitpck)
				REQUIRE(meta.value_desc.nbits == om.nbits);
			if (om.encoding == oroch::bitpfr) {
				REQUIRE(meta.noutliers == om.noutliers);
				REQUIRE(meta.outlier_index_desc.encoding == om.index_encoding);
				REQUIRE(meta.outlier_value_desc.encoding == om.value_encoding);
			}

			// Metadata bytes, data bytes (integer_codec::encode) and decode.
			std::vector<uint8_t> want(orc_group_bound(dt, n) + 64), got(want.size());
			uint8_t mbytes[64], wbytes[64];
			oroch::dst_bytes_t mp = mbytes;
			meta.encode(mp);
			const size_t wm = orc_meta_encode(dt, &om, wbytes);
			REQUIRE(size_t(mp - mbytes) == wm);
			REQUIRE(std::memcmp(mbytes, wbytes, wm) == 0);
			const size_t wd = orc_codec_encode(dt, v.data(), n, &om, want.data());
			oroch::dst_bytes_t dp = got.data();
			oroch::integer_codec<T>::encode(dp, v.begin(), v.end(), meta);
			REQUIRE(size_t(dp - got.data()) == wd);
			REQUIRE(std::memcmp(got.data(), want.data(), wd) == 0);
			std::vector<T> back(n);
			oroch::src_bytes_t sp = got.data();
			typename oroch::integer_codec<T>::metadata meta2;
			oroch::src_bytes_t mq = mbytes;
			meta2.decode(mq);
			oroch::integer_codec<T>::decode(back.begin(), back.end(), sp, meta2);
			REQUIRE(size_t(sp - got.data()) == wd);
			REQUIRE(back == v);

			// integer_group, aligned and not.
			const size_t wg = orc_group_encode(dt, v.data(), n, want.data());
			oroch::integer_group<T> group;
			group.encode(v.begin(), v.end());
			REQUIRE(group.size() == wg);
			REQUIRE(std::memcmp(group.data(), want.data(), wg) == 0);
			std::fill(back.begin(), back.end(), T(0));
			group.decode(back.begin(), back.end());
			REQUIRE(back == v);
			group.encode(v.begin(), v.end(), false);
			REQUIRE(group.size() == wm + wd);
			std::fill(back.begin(), back.end(), T(1));
			group.decode(back.begin(), back.end(), false);
			REQUIRE(back == v);
		}
	}
}

template <typename T>
static void test_sequences()
{
	const int dt = oroch::hip::dtype_of<T>();
	constexpr unsigned W = sizeof(T) * 8;
	using U = typename std::make_unsigned<T>::type;
	const size_t sizes[] = {1, 7, 130, 1000, 5000};
	for (size_t n : sizes) {
		for (unsigned nbits : {1u, 3u, 7u, W / 2, W - 1, W}) {
			std::vector<T> v(n);
			// Values whose zigzag image fits nbits.
			for (auto &x : v) {
				const U z = nbits == W ? U(rnd()) : U(rnd() & ((U(1) << nbits) - 1));
				x = oroch::zigzag_codec<T>::decode_if_signed(z);
			}
			using pck = oroch::bitpck_codec<T>;
			std::vector<uint8_t> want(pck::space(n, nbits)), got(want.size());
			REQUIRE(orc_bitpck_encode(dt, v.data(), n, nbits, want.data()) == want.size());
			oroch::dst_bytes_t d = got.data();
			pck::encode(d, v.begin(), v.end(), nbits);
			REQUIRE(size_t(d - got.data()) == want.size());
			REQUIRE(got == want);
			std::vector<T> back(n);
			oroch::src_bytes_t s = got.data();
			pck::decode(back.begin(), back.end(), s, nbits);
			REQUIRE(size_t(s - got.data()) == want.size());
			REQUIRE(back == v);

			// bitfor: origin anywhere, deltas fit nbits.
			const T origin = T(U(rnd()));
			for (auto &x : v) {
				const U delta = nbits == W ? U(rnd()) : U(rnd() & ((U(1) << nbits) - 1));
				x = T(U(U(origin) + delta));
			}
			using bfor = oroch::bitfor_codec<T>;
			typename bfor::parameters params(origin, nbits);
			REQUIRE(orc_bitfor_encode(dt, v.data(), n, oroch::hip::raw64<T>(origin), nbits, want.data())
				== want.size());
			d = got.data();
			bfor::encode(d, v.begin(), v.end(), params);
			REQUIRE(got == want);
			s = got.data();
			bfor::decode(back.begin(), back.end(), s, params);
			REQUIRE(back == v);

			// bitpfr: some deltas do not fit.
			if (nbits == W)
				continue;
			for (size_t i = 0; i < n; i++)
				if (rnd() % 16 == 0)
					v[i] = T(U(U(origin) + U(rnd())));
			using pfr = oroch::bitpfr_codec<T>;
			typename pfr::exceptions x;
			typename pfr::parameters pp(origin, nbits, x);
			std::vector<uint64_t> wi(n), wv(n);
			size_t wn = 0;
			orc_bitpfr_encode(dt, v.data(), n, oroch::hip::raw64<T>(origin), nbits, want.data(),
					  wi.data(), wv.data(), &wn);
			d = got.data();
			pfr::encode(d, v.begin(), v.end(), pp);
			REQUIRE(got == want);
			REQUIRE(x.index == n);
			REQUIRE(x.indices.size() == wn);
			bool same = x.indices.size() == wn;
			for (size_t i = 0; same && i < wn; i++)
				same = x.indices[i] == wi[i] && uint64_t(x.values[i]) == wv[i];
			REQUIRE(same);
			s = got.data();
			std::fill(back.begin(), back.end(), T(0));
			pfr::decode(back.begin(), back.end(), s, pp);
			REQUIRE(back == v);
			// The two halves of decode, as integer_codec.h:583-609 uses them.
			s = got.data();
			std::fill(back.begin(), back.end(), T(0));
			pfr::decode_basic(back.begin(), back.end(), s, pp);
			pfr::decode_patch(back.begin(), pp);
			REQUIRE(back == v);
		}
	}
}

// Block-level members of bitpck_codec (bitpck.h:73-225, 258-266), in the shape
// of tests/unit/bitblk.cc: values wider than nbits are truncated by the mask.
static void test_bitpck_blocks()
{
	using codec = oroch::bitpck_codec<uint32_t>;
	for (size_t nbits : {1, 5, 13, 31, 32}) {
		const size_t c = codec::capacity(nbits);
		std::vector<uint32_t> v(c + 3);
		for (auto &x : v)
			x = uint32_t(rnd());
		std::vector<uint8_t> bytes(32), want(32);
		oroch::zigzag_codec<uint32_t> vc;
		oroch::dst_bytes_t d = bytes.data();
		auto it = v.begin();
		codec::block_encode(d, it, v.end(), nbits, vc);
		REQUIRE(size_t(it - v.begin()) == c);
		codec::block_encode(d, it, v.end(), nbits, vc);
		REQUIRE(it == v.end() && d == bytes.data() + 32);
		orc_bitpck_encode(5, v.data(), v.size(), unsigned(nbits), want.data());
		REQUIRE(bytes == want);
		const uint32_t mask = nbits == 32 ? ~0u : (1u << nbits) - 1;
		std::vector<uint32_t> back(c + 3);
		oroch::src_bytes_t s = bytes.data();
		codec::block_decode(back.begin(), s, nbits, vc);
		codec::block_decode(back.begin() + c, back.end(), s, nbits, vc);
		REQUIRE(s == bytes.data() + 32);
		for (size_t i = 0; i < v.size(); i++)
			REQUIRE(back[i] == (v[i] & mask));
		REQUIRE(codec::fetch(bytes.data(), c + 1, nbits) == (v[c + 1] & mask));
		REQUIRE(codec::block_fetch(bytes.data(), c - 1, nbits, vc) == (v[c - 1] & mask));
	}
	// A 64-bit codec fed 32-bit data, tests/unit/bitpck.cc:59-68.
	std::array<uint32_t, 100> small;
	for (auto &x : small)
		x = uint32_t(rnd() & 0x1ffff);
	using wide = oroch::bitpck_codec<uint64_t>;
	std::vector<uint8_t> bytes(wide::space(small.size(), 17));
	oroch::dst_bytes_t d = bytes.data();
	wide::encode(d, small.begin(), small.end(), 17);
	std::array<uint32_t, 100> back;
	oroch::src_bytes_t s = bytes.data();
	wide::decode(back.begin(), back.end(), s, 17);
	REQUIRE(back == small);
	using bfor = oroch::bitfor_codec<int16_t>;
	std::array<int16_t, 40> w;
	for (size_t i = 0; i < w.size(); i++)
		w[i] = int16_t(-300 + int(i) * 3);
	bfor::parameters params(-300, 7);
	std::vector<uint8_t> fb(bfor::space(w.size(), 7));
	d = fb.data();
	bfor::encode(d, w.begin(), w.end(), params);
	for (size_t i = 0; i < w.size(); i++)
		REQUIRE(bfor::fetch(fb.data(), i, params) == w[i]);
}

// tests/unit/bitpfr.cc:13-30.
static void test_bitpfr_value_codec()
{
	using codec = oroch::bitpfr_codec<uint32_t>;
	codec::exceptions excpts;
	codec::parameters params(1000, 7, excpts);
	params.value_encode(1000);
	REQUIRE(excpts.index == 1);
	REQUIRE(excpts.indices.size() == 0);
	REQUIRE(excpts.values.size() == 0);
	params.value_encode(1000 + (1 << 7));
	REQUIRE(excpts.index == 2);
	REQUIRE(excpts.indices.size() == 1);
	REQUIRE(excpts.indices[0] == 1);
	REQUIRE(excpts.values.size() == 1);
	REQUIRE(excpts.values[0] == 1);
}

// bitpck.h:42: any class with value_encode / value_decode serves as the value
// codec.  One the kernels do not know is applied on the host; the packed bytes are
// those of its images.
template <typename T>
struct flip_codec
{
	using U = typename oroch::integer_traits<T>::unsigned_t;
	U value_encode(T v) const { return U(U(v) ^ U(0x155)); }
	T value_decode(U u) const { return T(U(u ^ U(0x155))); }
};

template <typename T>
static void test_bitpck_custom_value_codec()
{
	using U = typename oroch::integer_traits<T>::unsigned_t;
	using codec = oroch::bitpck_codec<T, flip_codec<T>>;
	using plain = oroch::bitpck_codec<U>;
	constexpr size_t N = 1000, BITS = 9;
	std::vector<T> values(N), back(N);
	std::vector<U> images(N);
	flip_codec<T> flip;
	for (size_t i = 0; i < N; i++) {
		values[i] = T(rnd() & ((1u << BITS) - 1));
		images[i] = flip.value_encode(values[i]);
	}
	std::vector<uint8_t> bytes(codec::space(N, BITS) + 16, 0xee), want(plain::space(N, BITS));
	oroch::dst_bytes_t d = bytes.data(), w = want.data();
	codec::encode(d, values.begin(), values.end(), BITS);
	plain::encode(w, images.begin(), images.end(), BITS);
	REQUIRE(size_t(d - bytes.data()) == want.size());
	REQUIRE(std::equal(want.begin(), want.end(), bytes.begin()));
	REQUIRE(bytes[want.size()] == 0xee);
	oroch::src_bytes_t s = bytes.data();
	codec::decode(back.begin(), back.end(), s, BITS);
	REQUIRE(size_t(s - bytes.data()) == want.size());
	REQUIRE(back == values);
	for (size_t i : {size_t(0), size_t(13), size_t(14), N - 1})
		REQUIRE(codec::fetch(bytes.data(), i, BITS) == values[i]);
	// one block at a time
	uint8_t block[16];
	d = block;
	auto it = values.begin();
	codec::block_encode(d, it, values.end(), BITS, flip);
	REQUIRE(size_t(it - values.begin()) == codec::capacity(BITS));
	REQUIRE(std::equal(block, block + 16, want.begin()));
	std::vector<T> first(codec::capacity(BITS));
	s = block;
	codec::block_decode(first.begin(), s, BITS, flip);
	REQUIRE(std::equal(first.begin(), first.end(), values.begin()));
}

// tests/unit/bitpfr.cc:32-92: 128 small values, 24 growing powers of two.
template <typename T>
static void test_bitpfr_reference_case(T fref)
{
	constexpr int BITS = 7, SMALL = 128, LARGE = 24, INTS = SMALL + LARGE;
	using codec = oroch::bitpfr_codec<T>;
	std::array<uint8_t, codec::basic_codec::space(INTS, BITS)> bytes;
	std::array<T, INTS> integers, integers2;
	typename codec::exceptions excpts;
	typename codec::parameters params(fref, BITS, excpts);
	for (int i = 0; i < SMALL; i++)
		integers[i] = T(i + fref);
	for (int i = SMALL; i < INTS; i++)
		integers[i] = T(1000 + (1 << (BITS + (i - SMALL))));
	oroch::dst_bytes_t d_it = bytes.data();
	codec::encode(d_it, integers.begin(), integers.end(), params);
	REQUIRE(excpts.indices.size() == LARGE);
	REQUIRE(excpts.values.size() == LARGE);
	oroch::src_bytes_t b_it = bytes.data();
	codec::decode(integers2.begin(), integers2.end(), b_it, params);
	for (int i = 0; i < INTS; i++)
		REQUIRE(integers2[i] == integers[i]);
}

// Two encode calls on one parameters object continue one index sequence
// (bitpfr.h:75-85 keeps the position in the exceptions object).
static void test_bitpfr_two_calls()
{
	using codec = oroch::bitpfr_codec<uint16_t>;
	std::vector<uint16_t> v(600);
	for (size_t i = 0; i < v.size(); i++)
		v[i] = uint16_t(50 + (i % 37 == 5 ? 4000 + i : i % 16));
	codec::exceptions one, two;
	codec::parameters p1(50, 4, one), p2(50, 4, two);
	std::vector<uint8_t> a(codec::basic_codec::space(600, 4)), b(codec::basic_codec::space(300, 4) * 2);
	oroch::dst_bytes_t d = a.data();
	codec::encode(d, v.begin(), v.end(), p1);
	d = b.data();
	codec::encode(d, v.begin(), v.begin() + 300, p2);
	codec::encode(d, v.begin() + 300, v.end(), p2);
	REQUIRE(one.index == 600 && two.index == 600);
	REQUIRE(one.indices == two.indices);
	REQUIRE(one.values == two.values);
}

// tests/unit/integer_group.cc:8-22 (fewer rounds: every call is a round trip
// to the device).
static void test_group_reference_case()
{
	oroch::integer_group<uint32_t> group;
	std::array<uint32_t, 8> integers, integers2;
	for (int t = 0; t < 200; t++) {
		for (auto &x : integers)
			x = uint32_t(rnd() & 0xfff);
		group.encode(integers.begin(), integers.end());
		group.decode(integers2.begin(), integers2.end());
		REQUIRE(integers2 == integers);
	}
}

template <typename T>
static void test_blocks()
{
	const int dt = oroch::hip::dtype_of<T>();
	for (size_t n : {size_t(0), size_t(1), size_t(256), size_t(257), size_t(70001)}) {
		std::vector<T> v;
		for (size_t b = 0; b * 256 < n; b++) {
			const std::vector<T> part = make_block<T>(int(rnd() % 8), std::min<size_t>(256, n - b * 256));
			v.insert(v.end(), part.begin(), part.end());
		}
		oroch::integer_blocks<T> blocks;
		blocks.encode(v.begin(), v.end());
		const size_t nb = (n + 255) / 256;
		std::vector<uint8_t> want(orc_group_bound(dt, 256) * (nb + 1) + 64);
		std::vector<uint64_t> offs(nb + 1);
		const size_t total = orc_blocks_encode(dt, v.data(), n, 256, 16, want.data(), offs.data());
		REQUIRE(blocks.stream().size() == total);
		REQUIRE(blocks.offsets() == offs);
		REQUIRE(std::memcmp(blocks.stream().data(), want.data(), total) == 0);
		std::vector<T> back(n);
		blocks.decode(back.begin(), back.end());
		REQUIRE(back == v);
	}
}

// The serialised form: block images equal to integer_group's, any block range
// decodable on its own, bytes survive a trip through a plain byte vector.
template <typename T>
static void test_container()
{
	const int dt = oroch::hip::dtype_of<T>();
	std::vector<T> v;
	for (int b = 0; b < 40; b++) {
		const std::vector<T> part = make_block<T>(int(rnd() % 8), b == 39 ? 100 : 256);
		v.insert(v.end(), part.begin(), part.end());
	}
	oroch::integer_container<T> c;
	c.encode(v.begin(), v.end());
	REQUIRE(c.nvalues() == v.size() && c.nblocks() == 40);
	std::vector<uint8_t> want(orc_group_bound(dt, 256) * 41 + 64);
	std::vector<uint64_t> offs(41);
	const size_t total = orc_blocks_encode(dt, v.data(), v.size(), 256, 16, want.data(), offs.data());
	const size_t at = (64 + 8 * 41 + 15) / 16 * 16;
	REQUIRE(c.bytes().size() == at + total);
	REQUIRE(std::memcmp(c.bytes().data() + at, want.data(), total) == 0);
	REQUIRE(std::memcmp(c.bytes().data() + 64, offs.data(), 8 * 41) == 0);
	oroch::integer_container<T> d(std::vector<uint8_t>(c.bytes()));
	std::vector<T> back(v.size());
	d.decode(back.begin(), back.end());
	REQUIRE(back == v);
	std::vector<T> some(3 * 256);
	d.decode_blocks(7, 3, some.begin());
	REQUIRE(std::equal(some.begin(), some.end(), v.begin() + 7 * 256));
	bool thrown = false;
	try {
		std::vector<uint8_t> cut(c.bytes().begin(), c.bytes().end() - 16);
		oroch::integer_container<T> e(std::move(cut));
	} catch (const std::runtime_error &) {
		thrown = true;
	}
	REQUIRE(thrown);
}

// tests/unit/integer_array.cc: "small integer array" and "large integer array".
static void test_array_reference_cases()
{
	{
		oroch::integer_array<int32_t> array;
		REQUIRE(array.find(0) == oroch::not_found);
		REQUIRE(array.find(1) == oroch::not_found);
		array.insert(0, 0);
		REQUIRE(array.at(0) == 0);
		REQUIRE(array.find(0) == 0);
		array.insert(0, 1);
		REQUIRE(array.at(0) == 1);
		REQUIRE(array.at(1) == 0);
		REQUIRE(array.find(0) == 1);
		REQUIRE(array.find(1) == 0);
	}
	const size_t n = 100000;
	oroch::integer_array<int32_t> array;
	for (size_t i = 0; i < n; i++)
		array.insert(i, int32_t(i));
	array.insert(0, -1);
	REQUIRE(array.size() == n + 1);
	// every 7th value on its own (one device pass per call), then all in one pass
	size_t bad = 0;
	for (size_t i = 0; i < n; i += 7)
		bad += array.find(int32_t(i)) != i + 1;
	REQUIRE(bad == 0);
	std::vector<int32_t> all(n);
	for (size_t i = 0; i < n; i++)
		all[i] = int32_t(i);
	const std::vector<size_t> where = array.find(all);
	for (size_t i = 0; i < n; i++)
		bad += where[i] != i + 1;
	REQUIRE(bad == 0);
	REQUIRE(array.find(-1) == 0);
	REQUIRE(array.find(int32_t(n)) == oroch::not_found);
}

// The array against a std::vector that sees the same operations; the groups
// against the oracle's block images.
template <typename T>
static void test_array()
{
	const int dt = oroch::hip::dtype_of<T>();
	std::vector<T> model;
	oroch::integer_array<T> array;
	REQUIRE(array.empty());
	const size_t counts[] = {3, 253, 256, 1, 700, 2048, 11};
	for (size_t count : counts) {
		std::vector<T> piece;
		while (piece.size() < count) {
			const std::vector<T> part = make_block<T>(int(rnd() % 8), std::min<size_t>(256, count - piece.size()));
			piece.insert(piece.end(), part.begin(), part.end());
		}
		array.append(piece.begin(), piece.end());
		model.insert(model.end(), piece.begin(), piece.end());
	}
	const size_t spots[] = {model.size(), model.size() - 1, 1000, 256, 255, 0};
	for (size_t spot : spots) {
		const T v = T(rnd());
		array.insert(spot, v);
		model.insert(model.begin() + spot, v);
	}
	REQUIRE(array.size() == model.size() && !array.empty());
	REQUIRE(array.read(0, model.size()) == model);
	REQUIRE(array.read(200, 600) == std::vector<T>(model.begin() + 200, model.begin() + 800));
	// groups: byte for byte what the oracle writes for the same values
	const size_t ngroups = model.size() / 256;
	std::vector<uint8_t> want(orc_group_bound(dt, 256) * (ngroups + 1) + 64), image(sizeof(T) * 256 + 64);
	std::vector<uint64_t> offs(ngroups + 1);
	orc_blocks_encode(dt, model.data(), ngroups * 256, 256, 16, want.data(), offs.data());
	size_t differ = 0;
	for (size_t g = 0; g < ngroups; g++) {
		uint64_t bytes = 0;
		oroch::hip::check(oroch_array_image(array.handle(), g, image.data(), image.size(), &bytes), "image");
		differ += bytes != offs[g + 1] - offs[g] || std::memcmp(image.data(), want.data() + offs[g], bytes) != 0;
	}
	REQUIRE(differ == 0);
	uint64_t bytes = 0;
	REQUIRE(oroch_array_image(array.handle(), ngroups, image.data(), image.size(), &bytes) == OROCH_E_RANGE);
	// at / find, one by one and in bulk, groups and tail
	std::vector<size_t> pos;
	std::vector<T> needles;
	for (int i = 0; i < 40; i++) {
		pos.push_back(size_t(rnd() % model.size()));
		needles.push_back(i % 4 == 3 ? T(rnd()) : model[pos.back()]);
	}
	pos.push_back(model.size() - 1);
	needles.push_back(model.back());
	const std::vector<T> got = array.at(pos);
	const std::vector<size_t> found = array.find(needles);
	for (size_t i = 0; i < pos.size(); i++) {
		REQUIRE(got[i] == model[pos[i]]);
		REQUIRE(array[pos[i]] == model[pos[i]]);
		const auto it = std::find(model.begin(), model.end(), needles[i]);
		const size_t expect = it == model.end() ? oroch::not_found : size_t(it - model.begin());
		REQUIRE(found[i] == expect);
		REQUIRE(array.find(needles[i]) == expect);
	}
	bool thrown = false;
	try {
		array.at(model.size());
	} catch (const std::out_of_range &) {
		thrown = true;
	}
	REQUIRE(thrown);
	thrown = false;
	try {
		array.insert(model.size() + 1, T(0));
	} catch (const std::out_of_range &) {
		thrown = true;
	}
	REQUIRE(thrown);
	// through a byte string and back; the bytes are the container of the same values
	{
		const std::vector<uint8_t> bytes = array.save();
		oroch::integer_container<T> c;
		c.encode(model.begin(), model.end());
		REQUIRE(bytes == c.bytes());
		oroch::integer_array<T> again;
		again.load(bytes);
		REQUIRE(again.size() == model.size());
		REQUIRE(again.read(0, model.size()) == model);
		again.insert(again.size(), T(5));
		REQUIRE(again.at(model.size()) == T(5));
	}
	std::ostringstream os;
	array.group_info(os);
	const std::string text = os.str();
	REQUIRE(size_t(std::count(text.begin(), text.end(), '\n')) == ngroups);
	array.clear();
	REQUIRE(array.empty() && array.find(T(1)) == oroch::not_found);
}

// tests/bench/group-bench.cc:8-30: one group of 1000 values (0..999), which the
// selection codes as bitfor or bitpck, encoded and decoded over and over.
static void test_group_bench_scenario()
{
	std::vector<uint32_t> integers(1000), integers2(1000);
	for (size_t i = 0; i < integers.size(); i++)
		integers[i] = uint32_t(i);
	std::vector<uint8_t> want(orc_group_bound(oroch::hip::dtype_of<uint32_t>(), integers.size()) + 64);
	const size_t wg = orc_group_encode(oroch::hip::dtype_of<uint32_t>(), integers.data(), integers.size(), want.data());
	oroch::integer_group<uint32_t> group;
	for (int round = 0; round < 5; round++) {
		group.encode(integers.begin(), integers.end());
		REQUIRE(group.size() == wg);
		REQUIRE(std::memcmp(group.data(), want.data(), wg) == 0);
		std::fill(integers2.begin(), integers2.end(), 0u);
		group.decode(integers2.begin(), integers2.end());
		REQUIRE(integers2 == integers);
	}
	oroch::integer_codec<uint32_t>::metadata meta;
	group.decode(meta);
	REQUIRE(meta.value_desc.encoding == oroch::bitfor || meta.value_desc.encoding == oroch::bitpck);
	REQUIRE(meta.value_desc.nbits == 10);
}

// integer_codec.h:386-393: encode codes the values the way the metadata it is
// handed says, whether or not select() would have chosen that.
template <typename T>
static void test_forced_metadata(size_t n)
{
	const int dt = oroch::hip::dtype_of<T>();
	constexpr int W = sizeof(T) * 8;
	const std::vector<T> v = make_block<T>(4, n);
	uint64_t mn, mx, hist[65];
	orc_stats(dt, v.data(), n, &mn, &mx, hist);
	unsigned range_bits = 0;
	for (int b = 0; b <= W; b++)
		if (hist[b])
			range_bits = unsigned(b);
	struct forced { uint32_t encoding, nbits; bool origin; };
	const forced cases[] = {{oroch::normal, 0, false}, {oroch::varint, 0, false}, {oroch::varfor, 0, true},
				{oroch::bitpck, uint32_t(W), false}, {oroch::bitfor, range_bits, true}};
	for (const forced &f : cases) {
		orc_meta om;
		std::memset(&om, 0, sizeof om);
		om.encoding = f.encoding;
		om.nbits = f.nbits;
		om.origin = f.origin ? mn : 0;
		typename oroch::integer_codec<T>::metadata meta;
		meta.value_desc.encoding = oroch::encoding_t(f.encoding);
		meta.value_desc.nbits = f.nbits;
		meta.value_desc.origin = f.origin ? T(typename std::make_unsigned<T>::type(mn)) : T(0);
		std::vector<uint8_t> want(n * 16 + 64), got(want.size());
		const size_t wd = orc_codec_encode(dt, v.data(), n, &om, want.data());
		oroch::dst_bytes_t dp = got.data();
		oroch::integer_codec<T>::encode(dp, v.begin(), v.end(), meta);
		REQUIRE(size_t(dp - got.data()) == wd);
		REQUIRE(std::memcmp(got.data(), want.data(), wd) == 0);
		std::vector<T> back(n);
		oroch::src_bytes_t sp = got.data();
		oroch::integer_codec<T>::decode(back.begin(), back.end(), sp, meta);
		REQUIRE(size_t(sp - got.data()) == wd);
		REQUIRE(back == v);
	}
}

static void test_errors()
{
	// More values than a block in one call: one group, as in the reference.
	std::vector<int32_t> v(257, 3);
	oroch::integer_codec<int32_t>::metadata meta;
	oroch::integer_codec<int32_t>::select(meta, v.begin(), v.end());
	REQUIRE(meta.value_desc.encoding == oroch::naught && meta.value_desc.origin == 3);
	// Empty input.
	std::vector<int32_t> w = {1, 2, 3, 4, 5, 6, 7, 8};
	oroch::integer_codec<int32_t>::select(meta, w.begin(), w.begin());
	REQUIRE(meta.value_desc.encoding == oroch::normal);
	REQUIRE(meta.dataspace() == 0 && meta.metaspace() == 1);
	// One value out of the middle of a packed sequence (bitpck.h:258-266).
	std::vector<uint32_t> p(1000);
	for (size_t i = 0; i < p.size(); i++)
		p[i] = uint32_t(rnd() & 0x1fff);
	std::vector<uint8_t> packed(oroch::bitpck_codec<uint32_t>::space(p.size(), 13));
	oroch::dst_bytes_t d = packed.data();
	oroch::bitpck_codec<uint32_t>::encode(d, p.begin(), p.end(), 13);
	for (size_t i : {size_t(0), size_t(8), size_t(9), size_t(500), size_t(999)})
		REQUIRE(oroch::bitpck_codec<uint32_t>::fetch(packed.data(), i, 13) == p[i]);
	oroch::bitfor_codec<int32_t>::parameters params(-700, 11);
	std::vector<int32_t> q(300);
	for (size_t i = 0; i < q.size(); i++)
		q[i] = int32_t(rnd() & 0x7ff) - 700;
	std::vector<uint8_t> framed(oroch::bitfor_codec<int32_t>::space(q.size(), 11));
	d = framed.data();
	oroch::bitfor_codec<int32_t>::encode(d, q.begin(), q.end(), params);
	for (size_t i : {size_t(0), size_t(11), size_t(150), size_t(299)})
		REQUIRE(oroch::bitfor_codec<int32_t>::fetch(framed.data(), i, params) == q[i]);
}

// varint_codec<T>::space / encode / decode over a sequence (the README's
// example, tests/unit/varint.cc:67-84 at scale): from the threshold on the
// calls go to the device and give the bytes of the scalar loop.
template <typename T>
static void test_varint_sequences()
{
	using codec = oroch::varint_codec<T>;
	const size_t n = oroch::hip::varint_device_threshold * 3 + 77;
	std::vector<T> values(n);
	for (size_t i = 0; i < n; i++) {
		const uint64_t r = rnd();
		// magnitudes of every length, both signs for signed T
		values[i] = T(r >> (8 + (r & 63) % (8 * sizeof(T))));
		if (std::numeric_limits<T>::is_signed && (r & 64))
			values[i] = T(~values[i]);
	}
	std::vector<uint8_t> want;
	for (size_t i = 0; i < n; i++) {
		uint8_t tmp[16];
		oroch::dst_bytes_t d = tmp;
		codec::value_encode(d, values[i]);
		want.insert(want.end(), tmp, d);
	}
	const size_t before = oroch::hip::varint_device_calls();
	REQUIRE(codec::space(values.begin(), values.end()) == want.size());
	std::vector<uint8_t> bytes(want.size() + 32, 0xee);
	oroch::dst_bytes_t d = bytes.data();
	codec::encode(d, values.data(), values.data() + n);
	REQUIRE(size_t(d - bytes.data()) == want.size());
	REQUIRE(std::equal(want.begin(), want.end(), bytes.begin()));
	REQUIRE(bytes[want.size()] == 0xee);
	std::vector<T> back(n);
	oroch::src_bytes_t s = bytes.data();
	codec::decode(back.begin(), back.end(), s);
	REQUIRE(size_t(s - bytes.data()) == want.size());
	REQUIRE(back == values);
	REQUIRE(oroch::hip::varint_device_calls() == before + 3);
	// ranges that are not arrays go through a copy
	{
		std::deque<T> in(values.begin(), values.end()), out(n);
		REQUIRE(codec::space(in.begin(), in.end()) == want.size());
		std::fill(bytes.begin(), bytes.end(), uint8_t(0xee));
		d = bytes.data();
		codec::encode(d, in.begin(), in.end());
		REQUIRE(size_t(d - bytes.data()) == want.size());
		REQUIRE(std::equal(want.begin(), want.end(), bytes.begin()));
		s = bytes.data();
		codec::decode(out.begin(), out.end(), s);
		REQUIRE(size_t(s - bytes.data()) == want.size());
		REQUIRE(std::equal(out.begin(), out.end(), values.begin()));
		// and raw pointers are written in place
		std::fill(back.begin(), back.end(), T(0));
		s = bytes.data();
		codec::decode(back.data(), back.data() + n, s);
		REQUIRE(back == values);
		REQUIRE(oroch::hip::varint_device_calls() == before + 7);
	}
	// below the threshold the loop runs on the host
	const size_t few = 100;
	REQUIRE(codec::space(values.begin(), values.begin() + few) > 0);
	d = bytes.data();
	codec::encode(d, values.begin(), values.begin() + few);
	s = bytes.data();
	codec::decode(back.begin(), back.begin() + few, s);
	REQUIRE(s == d && std::equal(back.begin(), back.begin() + few, values.begin()));
	REQUIRE(oroch::hip::varint_device_calls() == before + 7);
}

template <typename T>
static void run_type(const char *name)
{
	test_varint_sequences<T>();
	test_stats_select_group<T>();
	test_sequences<T>();
	test_blocks<T>();
	test_container<T>();
	test_array<T>();
	std::printf("%-8s ok so far: %d checks, %d failed\n", name, g_checks, g_failed);
	std::fflush(stdout);
}

int main()
{
	oroch::hip::check(oroch_hip_init(0), "oroch_hip_init");
	test_bitpck_blocks();
	test_bitpfr_value_codec();
	test_bitpck_custom_value_codec<int16_t>();
	test_bitpck_custom_value_codec<uint32_t>();
	test_bitpck_custom_value_codec<int64_t>();
	test_bitpfr_reference_case<uint32_t>(1000);
	test_bitpfr_reference_case<int32_t>(-1000);
	test_bitpfr_two_calls();
	test_group_reference_case();
	test_errors();
	test_group_bench_scenario();
	test_forced_metadata<int16_t>(100);
	test_forced_metadata<int32_t>(256);
	test_forced_metadata<uint32_t>(3000);
	test_forced_metadata<int64_t>(700);
	test_array_reference_cases();
	run_type<int8_t>("int8");
	run_type<uint8_t>("uint8");
	run_type<int16_t>("int16");
	run_type<uint16_t>("uint16");
	run_type<int32_t>("int32");
	run_type<uint32_t>("uint32");
	run_type<int64_t>("int64");
	run_type<uint64_t>("uint64");
	std::printf("%d checks, %d failed\n", g_checks, g_failed);
	return g_failed ? 1 : 0;
}
