/*
 * oroch_oracle.c -- CPU restatement of Oroch's block integer-compression path.
 *
 * TEST INFRASTRUCTURE ONLY (see oroch_oracle.h).  Every value is handled as a
 * raw W-bit pattern held in a uint64_t; arithmetic wraps modulo 2^W, which is
 * what the reference's promoted-int / signed-overflow arithmetic produces on
 * the compilers it is built with.  Signed comparisons sign-extend first.
 *
 * Citations are to files below /root/reference/oroch.
 */
#include "oroch_oracle.h"

#include <string.h>

typedef struct {
	unsigned W;    /* bits per element */
	int sgn;       /* signed element type */
	uint64_t mask; /* 2^W - 1 */
} ty_t;

static ty_t ty_of(int dtype)
{
	ty_t t;
	t.W = 8u << (dtype >> 1);
	t.sgn = !(dtype & 1);
	t.mask = t.W == 64 ? ~(uint64_t) 0 : (((uint64_t) 1 << t.W) - 1);
	return t;
}

/* Load / store element i as a raw little-endian W-bit pattern. */
static uint64_t ld(const void *p, size_t i, unsigned W)
{
	uint64_t v = 0;
	memcpy(&v, (const uint8_t *) p + i * (W / 8), W / 8);
	return v;
}

static void st(void *p, size_t i, unsigned W, uint64_t v)
{
	memcpy((uint8_t *) p + i * (W / 8), &v, W / 8);
}

/* Sign-extend a raw pattern for ordering comparisons. */
static int64_t sx(ty_t t, uint64_t raw)
{
	if (!t.sgn || t.W == 64)
		return (int64_t) raw;
	uint64_t s = (uint64_t) 1 << (t.W - 1);
	return (int64_t) ((raw ^ s) - s);
}

/* a < b in the element type's own ordering. */
static int lt(ty_t t, uint64_t a, uint64_t b)
{
	return t.sgn ? sx(t, a) < sx(t, b) : a < b;
}

/* integer_traits.h:166-169 -- bit length, 0 for 0. */
static unsigned used(uint64_t u)
{
	unsigned n = 0;
	while (u) {
		n++;
		u >>= 1;
	}
	return n;
}

/* zigzag.h:47-50, 57-69 -- zigzag for signed T, identity for unsigned T. */
static uint64_t zz(ty_t t, uint64_t raw)
{
	if (!t.sgn)
		return raw;
	uint64_t sign = (raw >> (t.W - 1)) & 1;
	return ((raw << 1) ^ (sign ? t.mask : 0)) & t.mask;
}

/* zigzag.h:52-55, 71-83 */
static uint64_t unzz(ty_t t, uint64_t u)
{
	if (!t.sgn)
		return u;
	return ((u >> 1) ^ (0 - (u & 1))) & t.mask;
}

/* varint.h:56-59 */
static size_t nbits_space(size_t nbits)
{
	return (nbits + 6) / 7;
}

/* varint.h:66-72 on an already value-encoded unsigned. */
size_t orc_varint_space_u64(uint64_t u)
{
	return u == 0 ? 1 : nbits_space(used(u));
}

/* varint.h:74-83 */
static uint8_t *varint_put(uint8_t *dst, uint64_t u)
{
	while (u >= 0x80) {
		*dst++ = (uint8_t) (u | 0x80);
		u >>= 7;
	}
	*dst++ = (uint8_t) u;
	return dst;
}

/*
 * varint.h:85-115.  The value is accumulated in the W-bit unsigned type, so
 * groups shifted past bit W-1 are dropped.
 */
static const uint8_t *varint_get(const uint8_t *src, uint64_t wmask, uint64_t *out)
{
	uint64_t value = 0;
	unsigned shift = 0;
	for (;;) {
		uint64_t byte = *src++;
		if (shift < 64)
			value |= (byte & 0x7f) << shift;
		shift += 7;
		if ((byte & 0x80) == 0)
			break;
	}
	*out = value & wmask;
	return src;
}

/* bitpck.h:54-57 */
size_t orc_bitpck_capacity(size_t nbits)
{
	return 128 / nbits;
}

/* bitpck.h:61-71 */
size_t orc_bitpck_space(size_t nvalues, size_t nbits)
{
	size_t c = orc_bitpck_capacity(nbits);
	return 16 * ((nvalues + c - 1) / c);
}

/*
 * bitpck.h:73-128 -- value k of a block occupies bits [k*nbits, (k+1)*nbits)
 * of the block read as a 128-bit little-endian integer; the value is
 * truncated to nbits; unused bits are zero.
 */
static void block_put(uint8_t *block, size_t k, unsigned nbits, uint64_t value)
{
	uint64_t q[2];
	memcpy(q, block, 16);
	if (nbits < 64)
		value &= ((uint64_t) 1 << nbits) - 1;
	size_t bit = k * nbits;
	size_t w = bit >> 6, s = bit & 63;
	q[w] |= value << s;
	if (s != 0 && s + nbits > 64)
		q[w + 1] |= value >> (64 - s);
	memcpy(block, q, 16);
}

/* bitpck.h:203-225 */
static uint64_t block_get(const uint8_t *block, size_t k, unsigned nbits)
{
	uint64_t q[2];
	memcpy(q, block, 16);
	size_t bit = k * nbits;
	size_t w = bit >> 6, s = bit & 63;
	uint64_t x = q[w] >> s;
	if (s != 0 && s + nbits > 64)
		x |= q[w + 1] << (64 - s);
	if (nbits < 64)
		x &= ((uint64_t) 1 << nbits) - 1;
	return x;
}

/* bitpck.h:227-236 over already value-encoded unsigned inputs u[0..n). */
static size_t pack_u64(const uint64_t *u, size_t n, unsigned nbits, uint8_t *dst)
{
	size_t space = n ? orc_bitpck_space(n, nbits) : 0;
	size_t c = nbits ? orc_bitpck_capacity(nbits) : 1;
	memset(dst, 0, space);
	for (size_t i = 0; i < n; i++)
		block_put(dst + 16 * (i / c), i % c, nbits, u[i]);
	return space;
}

/* ---- whole-sequence codecs ------------------------------------------- */

size_t orc_bitpck_encode(int dtype, const void *src, size_t n, unsigned nbits, uint8_t *dst)
{
	ty_t t = ty_of(dtype);
	size_t space = n ? orc_bitpck_space(n, nbits) : 0;
	size_t c = orc_bitpck_capacity(nbits);
	memset(dst, 0, space);
	for (size_t i = 0; i < n; i++)
		block_put(dst + 16 * (i / c), i % c, nbits, zz(t, ld(src, i, t.W)));
	return space;
}

size_t orc_bitpck_decode(int dtype, const uint8_t *src, size_t n, unsigned nbits, void *dst)
{
	ty_t t = ty_of(dtype);
	size_t c = orc_bitpck_capacity(nbits);
	for (size_t i = 0; i < n; i++)
		st(dst, i, t.W, unzz(t, block_get(src + 16 * (i / c), i % c, nbits) & t.mask));
	return n ? orc_bitpck_space(n, nbits) : 0;
}

/* origin.h:46-49 through bitfor.h:59-63 */
size_t orc_bitfor_encode(int dtype, const void *src, size_t n, uint64_t origin, unsigned nbits,
			 uint8_t *dst)
{
	ty_t t = ty_of(dtype);
	size_t space = n ? orc_bitpck_space(n, nbits) : 0;
	size_t c = orc_bitpck_capacity(nbits);
	memset(dst, 0, space);
	for (size_t i = 0; i < n; i++)
		block_put(dst + 16 * (i / c), i % c, nbits, (ld(src, i, t.W) - origin) & t.mask);
	return space;
}

/* origin.h:51-54 through bitfor.h:65-69 */
size_t orc_bitfor_decode(int dtype, const uint8_t *src, size_t n, uint64_t origin, unsigned nbits,
			 void *dst)
{
	ty_t t = ty_of(dtype);
	size_t c = orc_bitpck_capacity(nbits);
	for (size_t i = 0; i < n; i++) {
		uint64_t u = block_get(src + 16 * (i / c), i % c, nbits) & t.mask;
		st(dst, i, t.W, (u + origin) & t.mask);
	}
	return n ? orc_bitpck_space(n, nbits) : 0;
}

/* bitpfr.h:75-85 (exception bookkeeping), offset.h:55-60 (gap = i - prev - 1) */
size_t orc_bitpfr_encode(int dtype, const void *src, size_t n, uint64_t origin, unsigned nbits,
			 uint8_t *dst, uint64_t *indices, uint64_t *values, size_t *nexc)
{
	ty_t t = ty_of(dtype);
	size_t space = n ? orc_bitpck_space(n, nbits) : 0;
	size_t c = orc_bitpck_capacity(nbits);
	/* bitpfr.h:65 -- (1ul << n) - 1 converted to unsigned_t */
	uint64_t lowmask = ((nbits < 64 ? ((uint64_t) 1 << nbits) : 0) - 1) & t.mask;
	uint64_t next = 0; /* offset_codec<size_t,1,false>(0): origin_ */
	size_t k = 0;
	memset(dst, 0, space);
	for (size_t i = 0; i < n; i++) {
		uint64_t u = (ld(src, i, t.W) - origin) & t.mask;
		if ((u & ~lowmask) != 0) {
			indices[k] = (uint64_t) i - next;
			next = (uint64_t) i + 1;
			values[k] = u >> nbits;
			k++;
		}
		block_put(dst + 16 * (i / c), i % c, nbits, u);
	}
	*nexc = k;
	return space;
}

/* bitpfr.h:101-124, offset.h:62-67 */
size_t orc_bitpfr_decode(int dtype, const uint8_t *src, size_t n, uint64_t origin, unsigned nbits,
			 const uint64_t *indices, const uint64_t *values, size_t nexc, void *dst)
{
	ty_t t = ty_of(dtype);
	size_t space = orc_bitfor_decode(dtype, src, n, origin, nbits, dst);
	uint64_t next = 0;
	for (size_t k = 0; k < nexc; k++) {
		uint64_t idx = next + indices[k];
		next = idx + 1;
		uint64_t u = (ld(dst, idx, t.W) - origin) & t.mask;
		u |= (values[k] << nbits) & t.mask;
		st(dst, idx, t.W, (u + origin) & t.mask);
	}
	return space;
}

size_t orc_varint_encode(int dtype, const void *src, size_t n, uint8_t *dst)
{
	ty_t t = ty_of(dtype);
	uint8_t *p = dst;
	for (size_t i = 0; i < n; i++)
		p = varint_put(p, zz(t, ld(src, i, t.W)));
	return (size_t) (p - dst);
}

size_t orc_varint_decode(int dtype, const uint8_t *src, size_t n, void *dst)
{
	ty_t t = ty_of(dtype);
	const uint8_t *p = src;
	for (size_t i = 0; i < n; i++) {
		uint64_t u;
		p = varint_get(p, t.mask, &u);
		st(dst, i, t.W, unzz(t, u));
	}
	return (size_t) (p - src);
}

/* ---- statistics and selection ----------------------------------------- */

/* integer_stats.h:48-53, 88-95 (min/max), 74-80, 97-102 (histogram) */
void orc_stats(int dtype, const void *src, size_t n, uint64_t *min_raw, uint64_t *max_raw,
	       uint64_t *hist)
{
	ty_t t = ty_of(dtype);
	/* integer_stats.h:108-109: min starts at max(), max at min() of T */
	uint64_t mn = t.sgn ? (t.mask >> 1) : t.mask;
	uint64_t mx = t.sgn ? ((uint64_t) 1 << (t.W - 1)) : 0;
	for (size_t i = 0; i < n; i++) {
		uint64_t v = ld(src, i, t.W);
		if (lt(t, v, mn))
			mn = v;
		if (lt(t, mx, v))
			mx = v;
	}
	*min_raw = mn;
	*max_raw = mx;
	if (hist) {
		memset(hist, 0, 65 * sizeof *hist);
		for (size_t i = 0; i < n; i++)
			hist[used((ld(src, i, t.W) - mn) & t.mask)]++;
	}
}

typedef struct {
	uint32_t encoding;
	uint64_t dataspace, metaspace; /* metaspace excludes the encoding byte */
	uint64_t origin;
	uint32_t nbits;
} desc_t;

/* integer_codec.h:405-420 -- strictly smaller replaces */
static void compare(desc_t *d, uint32_t encoding, uint64_t metaspace, uint64_t dataspace,
		    uint64_t origin, uint32_t nbits)
{
	if (dataspace + metaspace < d->dataspace + d->metaspace) {
		d->encoding = encoding;
		d->dataspace = dataspace;
		d->metaspace = metaspace;
		d->origin = origin;
		d->nbits = nbits;
	}
}

void orc_select(int dtype, const void *src, size_t n, orc_meta *meta)
{
	ty_t t = ty_of(dtype);
	uint64_t hist[65];
	uint64_t mn, mx;
	desc_t d;

	memset(meta, 0, sizeof *meta);
	/* fresh descriptors: integer_codec.h:83-90 */
	memset(&d, 0, sizeof d);
	d.encoding = ORC_NORMAL;
	meta->index_encoding = ORC_NORMAL;
	meta->value_encoding = ORC_NORMAL;

	orc_stats(dtype, src, n, &mn, &mx, hist);

	/* integer_codec.h:247-253 */
	if (n == 0) {
		meta->encoding = ORC_NORMAL;
		meta->metaspace = 1;
		return;
	}
	/* integer_codec.h:255-264 */
	if (mn == mx) {
		meta->encoding = ORC_NAUGHT;
		meta->origin = mn;
		meta->metaspace = 1 + orc_varint_space_u64(zz(t, mn));
		return;
	}

	/* select_basic, integer_codec.h:422-496 */
	d.encoding = ORC_NORMAL;
	d.dataspace = n * (t.W / 8);
	{
		uint64_t umax;
		if (t.sgn) {
			uint64_t a = zz(t, mn), b = zz(t, mx);
			umax = a > b ? a : b;
		} else {
			umax = mx;
		}
		unsigned nb = used(umax);
		compare(&d, ORC_BITPCK, 1, orc_bitpck_space(n, nb), 0, nb);

		uint64_t range = (mx - mn) & t.mask;
		nb = used(range);
		compare(&d, ORC_BITFOR, 1 + orc_varint_space_u64(zz(t, mn)),
			orc_bitpck_space(n, nb), mn, nb);

		uint64_t vispace = 0, vfspace = 0;
		for (size_t i = 0; i < n; i++) {
			uint64_t v = ld(src, i, t.W);
			vispace += orc_varint_space_u64(zz(t, v));
			vfspace += orc_varint_space_u64((v - mn) & t.mask);
		}
		compare(&d, ORC_VARINT, 0, vispace, 0, 0);
		compare(&d, ORC_VARFOR, orc_varint_space_u64(zz(t, mn)), vfspace, mn, 0);
	}

	/* integer_codec.h:271-383 */
	if (n >= 5) {
		uint64_t basic_metaspace = 1 + orc_varint_space_u64(zz(t, mn));
		uint64_t range = (mx - mn) & t.mask;
		unsigned nbits_max = used(range);
		uint64_t noutliers = n - hist[0];
		for (unsigned nbits = 1; nbits < nbits_max; nbits++) {
			if (hist[nbits] == 0)
				continue;
			noutliers -= hist[nbits];

			uint64_t basic_dataspace = orc_bitpck_space(n, nbits);
			uint64_t extra_metaspace = 2 + orc_varint_space_u64(noutliers);
			uint64_t valpck = orc_bitpck_space(noutliers, nbits_max - nbits);
			uint64_t valvar = 0;
			for (unsigned nb = nbits + 1; nb <= nbits_max; nb++)
				valvar += nbits_space(nb - nbits) * hist[nb];

			uint32_t value_encoding;
			uint64_t value_dataspace;
			if (valpck < valvar) {
				value_encoding = ORC_BITPCK;
				value_dataspace = valpck;
			} else {
				value_encoding = ORC_VARINT;
				value_dataspace = valvar;
			}

			/* the estimate gate, integer_codec.h:327-335 */
			uint64_t indmin = orc_bitpck_space(noutliers, 1);
			if (indmin > n)
				indmin = n;
			uint64_t estimate = basic_metaspace + extra_metaspace + basic_dataspace
					    + value_dataspace + indmin;
			uint64_t selected = d.dataspace + d.metaspace;
			if (estimate >= selected)
				continue;

			/* integer_codec.h:338-351 */
			unsigned indnbits = 1;
			uint64_t indvar = 0, next = 0;
			for (size_t i = 0; i < n; i++) {
				uint64_t u = ((ld(src, i, t.W) - mn) & t.mask) >> nbits;
				if (u == 0)
					continue;
				uint64_t j = (uint64_t) i - next;
				next = (uint64_t) i + 1;
				unsigned inb = used(j);
				if (indnbits < inb)
					indnbits = inb;
				indvar += orc_varint_space_u64(j);
			}
			uint64_t indpck = orc_bitpck_space(noutliers, indnbits);

			uint32_t index_encoding;
			uint64_t index_dataspace;
			if (indpck < indvar) {
				index_encoding = ORC_BITPCK;
				index_dataspace = indpck;
			} else {
				index_encoding = ORC_VARINT;
				index_dataspace = indvar;
			}

			uint64_t required = basic_metaspace + extra_metaspace + basic_dataspace
					    + value_dataspace + index_dataspace;
			if (required < selected) {
				d.encoding = ORC_BITPFR;
				d.origin = mn;
				d.nbits = nbits;
				meta->noutliers = noutliers;
				meta->value_encoding = value_encoding;
				meta->value_nbits = nbits_max - nbits;
				meta->index_encoding = index_encoding;
				meta->index_nbits = indnbits;
				d.metaspace = basic_metaspace + extra_metaspace;
				d.dataspace = basic_dataspace + value_dataspace + index_dataspace;
			}
		}
	}

	meta->encoding = d.encoding;
	meta->nbits = d.nbits;
	meta->origin = d.origin;
	meta->dataspace = d.dataspace;
	meta->metaspace = 1 + d.metaspace;
}

/* integer_codec.h:128-151, 177-182, 196-204 */
size_t orc_meta_encode(int dtype, const orc_meta *m, uint8_t *dst)
{
	ty_t t = ty_of(dtype);
	uint8_t *p = dst;
	*p++ = (uint8_t) m->encoding;
	switch (m->encoding) {
	case ORC_NAUGHT:
	case ORC_VARFOR:
		p = varint_put(p, zz(t, m->origin));
		break;
	case ORC_NORMAL:
	case ORC_VARINT:
		break;
	case ORC_BITPFR:
	case ORC_BITFOR:
		p = varint_put(p, zz(t, m->origin));
		*p++ = (uint8_t) m->nbits;
		break;
	case ORC_BITPCK:
		*p++ = (uint8_t) m->nbits;
		break;
	}
	if (m->encoding == ORC_BITPFR) {
		p = varint_put(p, m->noutliers);
		*p++ = m->index_encoding == ORC_BITPCK ? (uint8_t) m->index_nbits : 0;
		*p++ = m->value_encoding == ORC_BITPCK ? (uint8_t) m->value_nbits : 0;
	}
	return (size_t) (p - dst);
}

/* integer_codec.h:153-175, 184-194, 206-214 */
size_t orc_meta_decode(int dtype, const uint8_t *src, orc_meta *m)
{
	ty_t t = ty_of(dtype);
	const uint8_t *p = src;
	uint64_t u;
	memset(m, 0, sizeof *m);
	m->index_encoding = ORC_NORMAL;
	m->value_encoding = ORC_NORMAL;
	m->encoding = *p++;
	switch (m->encoding) {
	case ORC_NAUGHT:
	case ORC_VARFOR:
		p = varint_get(p, t.mask, &u);
		m->origin = unzz(t, u);
		break;
	case ORC_NORMAL:
	case ORC_VARINT:
		break;
	case ORC_BITPFR:
	case ORC_BITFOR:
		p = varint_get(p, t.mask, &u);
		m->origin = unzz(t, u);
		m->nbits = *p++;
		break;
	case ORC_BITPCK:
		m->nbits = *p++;
		break;
	}
	if (m->encoding == ORC_BITPFR) {
		p = varint_get(p, ~(uint64_t) 0, &u);
		m->noutliers = u;
		uint8_t nb = *p++;
		if (nb == 0) {
			m->index_encoding = ORC_VARINT;
		} else {
			m->index_encoding = ORC_BITPCK;
			m->index_nbits = nb;
		}
		nb = *p++;
		if (nb == 0) {
			m->value_encoding = ORC_VARINT;
		} else {
			m->value_encoding = ORC_BITPCK;
			m->value_nbits = nb;
		}
	}
	return (size_t) (p - src);
}

/*
 * encode_basic on a list of unsigned 64-bit-held values whose element type
 * is unsigned (size_t indices, unsigned_t outlier values): only bitpck and
 * varint occur there (integer_codec.h:573-580, 498-528).
 */
static size_t list_encode(uint32_t encoding, uint32_t nbits, const uint64_t *u, size_t n,
			  uint8_t *dst)
{
	if (encoding == ORC_BITPCK)
		return pack_u64(u, n, nbits, dst);
	uint8_t *p = dst;
	for (size_t i = 0; i < n; i++)
		p = varint_put(p, u[i]);
	return (size_t) (p - dst);
}

static size_t list_decode(uint32_t encoding, uint32_t nbits, uint64_t wmask, const uint8_t *src,
			  size_t n, uint64_t *u)
{
	if (encoding == ORC_BITPCK) {
		size_t c = orc_bitpck_capacity(nbits);
		for (size_t i = 0; i < n; i++)
			u[i] = block_get(src + 16 * (i / c), i % c, nbits) & wmask;
		return n ? orc_bitpck_space(n, nbits) : 0;
	}
	const uint8_t *p = src;
	for (size_t i = 0; i < n; i++)
		p = varint_get(p, wmask, &u[i]);
	return (size_t) (p - src);
}

/* Scratch for exception lists; blocks handled here are small. */
#include <stdlib.h>

size_t orc_codec_encode(int dtype, const void *src, size_t n, const orc_meta *m, uint8_t *dst)
{
	ty_t t = ty_of(dtype);
	uint8_t *p = dst;
	switch (m->encoding) {
	case ORC_NAUGHT: /* naught.h:37-40 */
		return 0;
	case ORC_NORMAL: /* normal.h:42-49 */
		memcpy(dst, src, n * (t.W / 8));
		return n * (t.W / 8);
	case ORC_VARINT: /* varint.h:133-139 with zigzag */
		return orc_varint_encode(dtype, src, n, dst);
	case ORC_VARFOR: /* varint.h:133-139 with origin_codec */
		for (size_t i = 0; i < n; i++)
			p = varint_put(p, (ld(src, i, t.W) - m->origin) & t.mask);
		return (size_t) (p - dst);
	case ORC_BITPCK:
		return orc_bitpck_encode(dtype, src, n, m->nbits, dst);
	case ORC_BITFOR:
		return orc_bitfor_encode(dtype, src, n, m->origin, m->nbits, dst);
	case ORC_BITPFR: { /* integer_codec.h:562-581 */
		uint64_t *idx = malloc((n + 1) * sizeof *idx);
		uint64_t *val = malloc((n + 1) * sizeof *val);
		size_t k;
		p += orc_bitpfr_encode(dtype, src, n, m->origin, m->nbits, p, idx, val, &k);
		p += list_encode(m->index_encoding, m->index_nbits, idx, k, p);
		p += list_encode(m->value_encoding, m->value_nbits, val, k, p);
		free(idx);
		free(val);
		return (size_t) (p - dst);
	}
	}
	return 0;
}

size_t orc_codec_decode(int dtype, const uint8_t *src, size_t n, const orc_meta *m, void *dst)
{
	ty_t t = ty_of(dtype);
	const uint8_t *p = src;
	switch (m->encoding) {
	case ORC_NAUGHT: /* naught.h:42-47 */
		for (size_t i = 0; i < n; i++)
			st(dst, i, t.W, m->origin);
		return 0;
	case ORC_NORMAL: /* normal.h:51-58 */
		memcpy(dst, src, n * (t.W / 8));
		return n * (t.W / 8);
	case ORC_VARINT:
		return orc_varint_decode(dtype, src, n, dst);
	case ORC_VARFOR:
		for (size_t i = 0; i < n; i++) {
			uint64_t u;
			p = varint_get(p, t.mask, &u);
			st(dst, i, t.W, (u + m->origin) & t.mask);
		}
		return (size_t) (p - src);
	case ORC_BITPCK:
		return orc_bitpck_decode(dtype, src, n, m->nbits, dst);
	case ORC_BITFOR:
		return orc_bitfor_decode(dtype, src, n, m->origin, m->nbits, dst);
	case ORC_BITPFR: { /* integer_codec.h:583-609 */
		size_t k = m->noutliers;
		uint64_t *idx = malloc((k + 1) * sizeof *idx);
		uint64_t *val = malloc((k + 1) * sizeof *val);
		size_t low = n ? orc_bitpck_space(n, m->nbits) : 0;
		p += low;
		p += list_decode(m->index_encoding, m->index_nbits, ~(uint64_t) 0, p, k, idx);
		p += list_decode(m->value_encoding, m->value_nbits, t.mask, p, k, val);
		orc_bitpfr_decode(dtype, src, n, m->origin, m->nbits, idx, val, k, dst);
		free(idx);
		free(val);
		return (size_t) (p - src);
	}
	}
	return 0;
}

/* ---- integer_group framing -------------------------------------------- */

size_t orc_group_bound(int dtype, size_t n)
{
	ty_t t = ty_of(dtype);
	/* the selected encoding never exceeds normal (1 meta byte + raw copy),
	 * except the constant case whose metadata is at most 1 + 10 bytes */
	size_t raw = 8 + n * (t.W / 8);
	return raw < 16 ? 16 : raw;
}

size_t orc_group_encode(int dtype, const void *src, size_t n, uint8_t *dst)
{
	orc_meta m;
	orc_select(dtype, src, n, &m);
	size_t off = orc_meta_encode(dtype, &m, dst);
	size_t data = (off + 7) & ~(size_t) 7; /* integer_group.h:50-52 */
	memset(dst + off, 0, data - off);
	size_t len = orc_codec_encode(dtype, src, n, &m, dst + data);
	return data + len;
}

size_t orc_group_decode(int dtype, const uint8_t *src, size_t n, void *dst)
{
	orc_meta m;
	size_t off = orc_meta_decode(dtype, src, &m);
	size_t data = (off + 7) & ~(size_t) 7; /* integer_group.h:70-72 */
	return data + orc_codec_decode(dtype, src + data, n, &m, dst);
}

size_t orc_blocks_encode(int dtype, const void *src, size_t n, size_t block, size_t align,
			 uint8_t *dst, uint64_t *offsets)
{
	ty_t t = ty_of(dtype);
	size_t nblocks = (n + block - 1) / block;
	size_t pos = 0;
	for (size_t b = 0; b < nblocks; b++) {
		size_t cnt = n - b * block < block ? n - b * block : block;
		offsets[b] = pos;
		size_t len = orc_group_encode(
			dtype, (const uint8_t *) src + b * block * (t.W / 8), cnt, dst + pos);
		size_t end = (pos + len + align - 1) / align * align;
		memset(dst + pos + len, 0, end - (pos + len));
		pos = end;
	}
	offsets[nblocks] = pos;
	return pos;
}

void orc_blocks_decode(int dtype, const uint8_t *src, const uint64_t *offsets, size_t n,
		       size_t block, void *dst)
{
	ty_t t = ty_of(dtype);
	size_t nblocks = (n + block - 1) / block;
	for (size_t b = 0; b < nblocks; b++) {
		size_t cnt = n - b * block < block ? n - b * block : block;
		orc_group_decode(dtype, src + offsets[b], cnt,
				 (uint8_t *) dst + b * block * (t.W / 8));
	}
}
