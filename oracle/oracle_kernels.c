/*
 * oracle_kernels.c -- CPU restatement of pgvector's scalar/SIMD distance
 * kernels and their SQL-callable wrappers.  TEST INFRASTRUCTURE ONLY (see
 * pgv_oracle.h).  Compiled with the reference's flags (Makefile:30) so that
 * gcc's auto-vectoriser is licensed to reassociate exactly as it is for the
 * reference build.
 */
#include "pgv_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#define ORA_X86 1
#endif

static __thread char ora_errbuf[160];

const char *
ora_last_error(void)
{
	return ora_errbuf;
}

static int
ora_check_dims(const char *type, int da, int db)
{
	/* CheckDims, src/vector.c:70-77; src/halfvec.c uses "halfvec" in the text */
	if (da != db)
	{
		snprintf(ora_errbuf, sizeof(ora_errbuf), "different %s dimensions %d and %d", type, da, db);
		return ORA_ERR_DIMS;
	}
	return ORA_OK;
}

/* ------------------------------------------------------------------ fp32 */

/* src/vector.c:560-574 */
float
ora_vector_l2_squared(int dim, const float *ax, const float *bx)
{
	float		acc = 0.0f;

	for (int i = 0; i < dim; i++)
	{
		float		d = ax[i] - bx[i];

		acc += d * d;
	}
	return acc;
}

/* src/vector.c:607-617 */
float
ora_vector_inner_product(int dim, const float *ax, const float *bx)
{
	float		acc = 0.0f;

	for (int i = 0; i < dim; i++)
		acc += ax[i] * bx[i];
	return acc;
}

/* src/vector.c:649-666: three fp32 accumulators, final divide in double */
double
ora_vector_cosine_similarity(int dim, const float *ax, const float *bx)
{
	float		sim = 0.0f,
				na = 0.0f,
				nb = 0.0f;

	for (int i = 0; i < dim; i++)
	{
		sim += ax[i] * bx[i];
		na += ax[i] * ax[i];
		nb += bx[i] * bx[i];
	}
	return (double) sim / sqrt((double) na * (double) nb);
}

/* src/vector.c:725-735 */
float
ora_vector_l1(int dim, const float *ax, const float *bx)
{
	float		acc = 0.0f;

	for (int i = 0; i < dim; i++)
		acc += fabsf(ax[i] - bx[i]);
	return acc;
}

/* ------------------------------------------------------------------ fp16 */

/* src/halfutils.h:62-141 (portable branch; bit-identical to _cvtsh_ss) */
float
ora_half_to_float(ora_half h)
{
	uint32_t	sign = ((uint32_t) h & 0x8000u) << 16;
	uint32_t	e = (h >> 10) & 0x1F;
	uint32_t	m = h & 0x3FF;
	uint32_t	bits;
	float		f;

	if (e == 31)
		bits = sign | (m == 0 ? 0x7F800000u : 0x7FC00000u | (m << 13));
	else if (e == 0)
	{
		if (m == 0)
			bits = sign;
		else
		{
			/* subnormal half: renormalise */
			int			ex = -14;

			while ((m & 0x400) == 0)
			{
				m <<= 1;
				ex--;
			}
			m &= 0x3FF;
			bits = sign | ((uint32_t) (ex + 127) << 23) | (m << 13);
		}
	}
	else
		bits = sign | ((e - 15 + 127) << 23) | (m << 13);
	memcpy(&f, &bits, 4);
	return f;
}

/* src/halfutils.h:146-233 (Float4ToHalfUnchecked: round-to-nearest-even,
 * overflow -> inf, values below half the smallest subnormal -> signed 0) */
ora_half
ora_float_to_half(float f)
{
	uint32_t	bin;
	uint16_t	res;
	int			ex,
				man;

	memcpy(&bin, &f, 4);
	ex = (bin & 0x7F800000u) >> 23;
	man = bin & 0x007FFFFF;
	res = (bin & 0x80000000u) >> 16;

	if (isinf(f))
		res |= 0x7C00;
	else if (isnan(f))
		res |= 0x7E00 | (man >> 13);
	else if (ex > 98)
	{
		int			sticky = man & 0xFFF;
		int			m,
					gr;

		ex -= 127;
		if (ex < -14)
		{
			int			shift = -ex - 14;

			man >>= shift;
			man += 1 << (23 - shift);
			sticky |= man & 0xFFF;
		}
		m = man >> 13;
		gr = (man >> 12) % 4;
		if (gr == 3 || (gr == 1 && sticky != 0))
			m++;
		if (m == 1024)
		{
			m = 0;
			ex++;
		}
		if (ex > 15)
			res |= 0x7C00;
		else
		{
			if (ex >= -14)
				res |= (ex + 15) << 10;
			res |= m;
		}
	}
	return res;
}

/* src/halfutils.c:29-43 */
float
ora_halfvec_l2_squared_default(int dim, const ora_half *ax, const ora_half *bx)
{
	float		acc = 0.0f;

	for (int i = 0; i < dim; i++)
	{
		float		d = ora_half_to_float(ax[i]) - ora_half_to_float(bx[i]);

		acc += d * d;
	}
	return acc;
}

/* src/halfutils.c:81-91 */
float
ora_halfvec_inner_product_default(int dim, const ora_half *ax, const ora_half *bx)
{
	float		acc = 0.0f;

	for (int i = 0; i < dim; i++)
		acc += ora_half_to_float(ax[i]) * ora_half_to_float(bx[i]);
	return acc;
}

/* src/halfutils.c:124-144 */
static double
halfvec_cosine_default(int dim, const ora_half *ax, const ora_half *bx)
{
	float		sim = 0.0f,
				na = 0.0f,
				nb = 0.0f;

	for (int i = 0; i < dim; i++)
	{
		float		a = ora_half_to_float(ax[i]);
		float		b = ora_half_to_float(bx[i]);

		sim += a * b;
		na += a * a;
		nb += b * b;
	}
	return (double) sim / sqrt((double) na * (double) nb);
}

/* src/halfutils.c:197-207 */
static float
halfvec_l1_default(int dim, const ora_half *ax, const ora_half *bx)
{
	float		acc = 0.0f;

	for (int i = 0; i < dim; i++)
		acc += fabsf(ora_half_to_float(ax[i]) - ora_half_to_float(bx[i]));
	return acc;
}

#ifdef ORA_X86
#define ORA_F16C __attribute__((target("avx,f16c,fma")))

/* horizontal sum in the reference's order s[0]+s[1]+...+s[7] (src/halfutils.c:66-68) */
ORA_F16C static inline float
hsum8_in_order(__m256 v)
{
	float		s[8];

	_mm256_storeu_ps(s, v);
	return s[0] + s[1] + s[2] + s[3] + s[4] + s[5] + s[6] + s[7];
}

/* src/halfutils.c:46-78: 8 lanes, one accumulator, scalar tail */
ORA_F16C static float
halfvec_l2_squared_f16c(int dim, const ora_half *ax, const ora_half *bx)
{
	int			body = dim & ~7;
	__m256		acc = _mm256_setzero_ps();
	float		r;
	int			i;

	for (i = 0; i < body; i += 8)
	{
		__m256		a = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (ax + i)));
		__m256		b = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (bx + i)));
		__m256		d = _mm256_sub_ps(a, b);

		acc = _mm256_fmadd_ps(d, d, acc);
	}
	r = hsum8_in_order(acc);
	for (; i < dim; i++)
	{
		float		d = ora_half_to_float(ax[i]) - ora_half_to_float(bx[i]);

		r += d * d;
	}
	return r;
}

/* src/halfutils.c:94-121 */
ORA_F16C static float
halfvec_inner_product_f16c(int dim, const ora_half *ax, const ora_half *bx)
{
	int			body = dim & ~7;
	__m256		acc = _mm256_setzero_ps();
	float		r;
	int			i;

	for (i = 0; i < body; i += 8)
	{
		__m256		a = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (ax + i)));
		__m256		b = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (bx + i)));

		acc = _mm256_fmadd_ps(a, b, acc);
	}
	r = hsum8_in_order(acc);
	for (; i < dim; i++)
		r += ora_half_to_float(ax[i]) * ora_half_to_float(bx[i]);
	return r;
}

/* src/halfutils.c:147-194 */
ORA_F16C static double
halfvec_cosine_f16c(int dim, const ora_half *ax, const ora_half *bx)
{
	int			body = dim & ~7;
	__m256		vs = _mm256_setzero_ps(),
				va = _mm256_setzero_ps(),
				vb = _mm256_setzero_ps();
	float		sim,
				na,
				nb;
	int			i;

	for (i = 0; i < body; i += 8)
	{
		__m256		a = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (ax + i)));
		__m256		b = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (bx + i)));

		vs = _mm256_fmadd_ps(a, b, vs);
		va = _mm256_fmadd_ps(a, a, va);
		vb = _mm256_fmadd_ps(b, b, vb);
	}
	sim = hsum8_in_order(vs);
	na = hsum8_in_order(va);
	nb = hsum8_in_order(vb);
	for (; i < dim; i++)
	{
		float		a = ora_half_to_float(ax[i]);
		float		b = ora_half_to_float(bx[i]);

		sim += a * b;
		na += a * a;
		nb += b * b;
	}
	return (double) sim / sqrt((double) na * (double) nb);
}

/* src/halfutils.c:211-239 */
ORA_F16C static float
halfvec_l1_f16c(int dim, const ora_half *ax, const ora_half *bx)
{
	int			body = dim & ~7;
	__m256		acc = _mm256_setzero_ps();
	const __m256 signbit = _mm256_set1_ps(-0.0f);
	float		r;
	int			i;

	for (i = 0; i < body; i += 8)
	{
		__m256		a = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (ax + i)));
		__m256		b = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (bx + i)));

		acc = _mm256_add_ps(acc, _mm256_andnot_ps(signbit, _mm256_sub_ps(a, b)));
	}
	r = hsum8_in_order(acc);
	for (; i < dim; i++)
		r += fabsf(ora_half_to_float(ax[i]) - ora_half_to_float(bx[i]));
	return r;
}
#endif							/* ORA_X86 */

/* HalfvecInit dispatch, src/halfutils.c:278-300 */
int
ora_halfvec_uses_f16c(void)
{
#ifdef ORA_X86
	static int	cached = -1;

	if (cached < 0)
		cached = __builtin_cpu_supports("avx") && __builtin_cpu_supports("f16c") && __builtin_cpu_supports("fma");
	return cached;
#else
	return 0;
#endif
}

float
ora_halfvec_l2_squared(int dim, const ora_half *ax, const ora_half *bx)
{
#ifdef ORA_X86
	if (ora_halfvec_uses_f16c())
		return halfvec_l2_squared_f16c(dim, ax, bx);
#endif
	return ora_halfvec_l2_squared_default(dim, ax, bx);
}

float
ora_halfvec_inner_product(int dim, const ora_half *ax, const ora_half *bx)
{
#ifdef ORA_X86
	if (ora_halfvec_uses_f16c())
		return halfvec_inner_product_f16c(dim, ax, bx);
#endif
	return ora_halfvec_inner_product_default(dim, ax, bx);
}

double
ora_halfvec_cosine_similarity(int dim, const ora_half *ax, const ora_half *bx)
{
#ifdef ORA_X86
	if (ora_halfvec_uses_f16c())
		return halfvec_cosine_f16c(dim, ax, bx);
#endif
	return halfvec_cosine_default(dim, ax, bx);
}

float
ora_halfvec_l1(int dim, const ora_half *ax, const ora_half *bx)
{
#ifdef ORA_X86
	if (ora_halfvec_uses_f16c())
		return halfvec_l1_f16c(dim, ax, bx);
#endif
	return halfvec_l1_default(dim, ax, bx);
}

/* ---------------------------------------------------------- SQL wrappers */

int
ora_l2_distance(int da, const float *a, int db, const float *b, double *out)
{
	int			rc = ora_check_dims("vector", da, db);

	if (rc)
		return rc;
	*out = sqrt((double) ora_vector_l2_squared(da, a, b));	/* vector.c:588 */
	return ORA_OK;
}

int
ora_l2_squared_distance(int da, const float *a, int db, const float *b, double *out)
{
	int			rc = ora_check_dims("vector", da, db);

	if (rc)
		return rc;
	*out = (double) ora_vector_l2_squared(da, a, b);	/* vector.c:604 */
	return ORA_OK;
}

int
ora_inner_product(int da, const float *a, int db, const float *b, double *out)
{
	int			rc = ora_check_dims("vector", da, db);

	if (rc)
		return rc;
	*out = (double) ora_vector_inner_product(da, a, b); /* vector.c:631 */
	return ORA_OK;
}

int
ora_negative_inner_product(int da, const float *a, int db, const float *b, double *out)
{
	int			rc = ora_check_dims("vector", da, db);

	if (rc)
		return rc;
	*out = (double) -ora_vector_inner_product(da, a, b);	/* vector.c:646 */
	return ORA_OK;
}

static double
clamp_cosine(double similarity)
{
	/* vector.c:689-695: NaN passes through both comparisons */
	if (similarity > 1)
		similarity = 1.0;
	else if (similarity < -1)
		similarity = -1.0;
	return 1.0 - similarity;
}

int
ora_cosine_distance(int da, const float *a, int db, const float *b, double *out)
{
	int			rc = ora_check_dims("vector", da, db);

	if (rc)
		return rc;
	*out = clamp_cosine(ora_vector_cosine_similarity(da, a, b));
	return ORA_OK;
}

static double
spherical_from_ip(double ip)
{
	/* vector.c:713-721 */
	if (ip > 1)
		ip = 1;
	else if (ip < -1)
		ip = -1;
	return acos(ip) / M_PI;
}

int
ora_spherical_distance(int da, const float *a, int db, const float *b, double *out)
{
	int			rc = ora_check_dims("vector", da, db);

	if (rc)
		return rc;
	*out = spherical_from_ip((double) ora_vector_inner_product(da, a, b));
	return ORA_OK;
}

int
ora_l1_distance(int da, const float *a, int db, const float *b, double *out)
{
	int			rc = ora_check_dims("vector", da, db);

	if (rc)
		return rc;
	*out = (double) ora_vector_l1(da, a, b);	/* vector.c:749 */
	return ORA_OK;
}

/* vector.c:767-780: double accumulate */
double
ora_vector_norm(int dim, const float *a)
{
	double		norm = 0.0;

	for (int i = 0; i < dim; i++)
		norm += (double) a[i] * (double) a[i];
	return sqrt(norm);
}

/* vector.c:785-819: zero vector stays zero; overflow of an element is an error */
int
ora_l2_normalize(int dim, const float *a, float *out)
{
	double		norm = ora_vector_norm(dim, a);

	for (int i = 0; i < dim; i++)
		out[i] = 0.0f;
	if (norm > 0)
	{
		for (int i = 0; i < dim; i++)
			out[i] = (float) (a[i] / norm);
		for (int i = 0; i < dim; i++)
			if (isinf(out[i]))
			{
				snprintf(ora_errbuf, sizeof(ora_errbuf), "value out of range: overflow");
				return ORA_ERR_OVERFLOW;
			}
	}
	return ORA_OK;
}

int
ora_halfvec_l2_distance(int da, const ora_half *a, int db, const ora_half *b, double *out)
{
	int			rc = ora_check_dims("halfvec", da, db);

	if (rc)
		return rc;
	*out = sqrt((double) ora_halfvec_l2_squared(da, a, b)); /* halfvec.c:584 */
	return ORA_OK;
}

int
ora_halfvec_l2_squared_distance(int da, const ora_half *a, int db, const ora_half *b, double *out)
{
	int			rc = ora_check_dims("halfvec", da, db);

	if (rc)
		return rc;
	*out = (double) ora_halfvec_l2_squared(da, a, b);
	return ORA_OK;
}

int
ora_halfvec_inner_product_f8(int da, const ora_half *a, int db, const ora_half *b, double *out)
{
	int			rc = ora_check_dims("halfvec", da, db);

	if (rc)
		return rc;
	*out = (double) ora_halfvec_inner_product(da, a, b);
	return ORA_OK;
}

int
ora_halfvec_negative_inner_product(int da, const ora_half *a, int db, const ora_half *b, double *out)
{
	int			rc = ora_check_dims("halfvec", da, db);

	if (rc)
		return rc;
	*out = (double) -ora_halfvec_inner_product(da, a, b);
	return ORA_OK;
}

int
ora_halfvec_cosine_distance(int da, const ora_half *a, int db, const ora_half *b, double *out)
{
	int			rc = ora_check_dims("halfvec", da, db);

	if (rc)
		return rc;
	*out = clamp_cosine(ora_halfvec_cosine_similarity(da, a, b));
	return ORA_OK;
}

int
ora_halfvec_spherical_distance(int da, const ora_half *a, int db, const ora_half *b, double *out)
{
	int			rc = ora_check_dims("halfvec", da, db);

	if (rc)
		return rc;
	*out = spherical_from_ip((double) ora_halfvec_inner_product(da, a, b));
	return ORA_OK;
}

int
ora_halfvec_l1_distance(int da, const ora_half *a, int db, const ora_half *b, double *out)
{
	int			rc = ora_check_dims("halfvec", da, db);

	if (rc)
		return rc;
	*out = (double) ora_halfvec_l1(da, a, b);
	return ORA_OK;
}

/* halfvec.c:703-719 */
double
ora_halfvec_l2_norm(int dim, const ora_half *a)
{
	double		norm = 0.0;

	for (int i = 0; i < dim; i++)
	{
		double		v = (double) ora_half_to_float(a[i]);

		norm += v * v;
	}
	return sqrt(norm);
}

/* halfvec.c:724-759 */
int
ora_halfvec_l2_normalize(int dim, const ora_half *a, ora_half *out)
{
	double		norm = ora_halfvec_l2_norm(dim, a);

	for (int i = 0; i < dim; i++)
		out[i] = 0;
	if (norm > 0)
	{
		for (int i = 0; i < dim; i++)
			out[i] = ora_float_to_half((float) (ora_half_to_float(a[i]) / norm));
		for (int i = 0; i < dim; i++)
			if ((out[i] & 0x7FFF) == 0x7C00)
			{
				snprintf(ora_errbuf, sizeof(ora_errbuf), "value out of range: overflow");
				return ORA_ERR_OVERFLOW;
			}
	}
	return ORA_OK;
}

/* ------------------------------------------------ opclass support functions */

/* FUNCTION 1: sql/vector.sql:409,415,422 (vector), :822,828,835 (halfvec);
 * hnsw l1: sql/vector.sql:445 */
double
ora_index_distance(int ops, int dtype, int dim, const void *a, const void *b)
{
	if (dtype == ORA_F32)
	{
		if (ops == ORA_OPS_L2)
			return (double) ora_vector_l2_squared(dim, a, b);
		if (ops == ORA_OPS_L1)
			return (double) ora_vector_l1(dim, a, b);
		return (double) -ora_vector_inner_product(dim, a, b);
	}
	if (ops == ORA_OPS_L2)
		return (double) ora_halfvec_l2_squared(dim, a, b);
	if (ops == ORA_OPS_L1)
		return (double) ora_halfvec_l1(dim, a, b);
	return (double) -ora_halfvec_inner_product(dim, a, b);
}

/* FUNCTION 3: sql/vector.sql:410,417,424 */
double
ora_kmeans_distance(int ops, int dtype, int dim, const void *a, const void *b)
{
	if (dtype == ORA_F32)
	{
		if (ops == ORA_OPS_L2)
			return sqrt((double) ora_vector_l2_squared(dim, a, b));
		return spherical_from_ip((double) ora_vector_inner_product(dim, a, b));
	}
	if (ops == ORA_OPS_L2)
		return sqrt((double) ora_halfvec_l2_squared(dim, a, b));
	return spherical_from_ip((double) ora_halfvec_inner_product(dim, a, b));
}

/* ------------------------------------------------------------------ bit */

/* the byte popcount table PostgreSQL exports as pg_number_of_ones (built, not copied) */
static unsigned char ones_of_byte[256];
static int	ones_ready;

static void
ones_init(void)
{
	if (ones_ready)
		return;
	for (int i = 0; i < 256; i++)
	{
		int			n = 0;

		for (int b = i; b; b >>= 1)
			n += b & 1;
		ones_of_byte[i] = (unsigned char) n;
	}
	ones_ready = 1;
}

/* BitHammingDistanceDefault, src/bitutils.c:49-73: 8 bytes at a time, then the byte tail */
uint64_t
ora_bit_hamming(uint32_t bytes, const unsigned char *ax, const unsigned char *bx)
{
	uint64_t	distance = 0;

	ones_init();
	for (; bytes >= sizeof(uint64_t); bytes -= sizeof(uint64_t))
	{
		uint64_t	axs,
					bxs;

		memcpy(&axs, ax, sizeof(uint64_t));
		memcpy(&bxs, bx, sizeof(uint64_t));
		distance += (uint64_t) __builtin_popcountll(axs ^ bxs);
		ax += sizeof(uint64_t);
		bx += sizeof(uint64_t);
	}
	for (uint32_t i = 0; i < bytes; i++)
		distance += ones_of_byte[ax[i] ^ bx[i]];
	return distance;
}

/* BitJaccardDistanceDefault, src/bitutils.c:99-131 */
double
ora_bit_jaccard(uint32_t bytes, const unsigned char *ax, const unsigned char *bx)
{
	uint64_t	ab = 0,
				aa = 0,
				bb = 0;

	ones_init();
	for (; bytes >= sizeof(uint64_t); bytes -= sizeof(uint64_t))
	{
		uint64_t	axs,
					bxs;

		memcpy(&axs, ax, sizeof(uint64_t));
		memcpy(&bxs, bx, sizeof(uint64_t));
		ab += (uint64_t) __builtin_popcountll(axs & bxs);
		aa += (uint64_t) __builtin_popcountll(axs);
		bb += (uint64_t) __builtin_popcountll(bxs);
		ax += sizeof(uint64_t);
		bx += sizeof(uint64_t);
	}
	for (uint32_t i = 0; i < bytes; i++)
	{
		ab += ones_of_byte[ax[i] & bx[i]];
		aa += ones_of_byte[ax[i]];
		bb += ones_of_byte[bx[i]];
	}
	if (ab == 0)
		return 1;
	return 1 - ((double) ab / (double) (aa + bb - ab));
}

/* CheckDims, src/bitvec.c:32-39 */
static int
ora_check_bits(int bits_a, int bits_b)
{
	if (bits_a != bits_b)
	{
		snprintf(ora_errbuf, sizeof(ora_errbuf), "different bit lengths %u and %u", (unsigned) bits_a, (unsigned) bits_b);
		return ORA_ERR_DIMS;
	}
	return ORA_OK;
}

/* hamming_distance, src/bitvec.c:45-55; VARBITBYTES = (bits + 7) / 8 */
int
ora_hamming_distance(int bits_a, const unsigned char *a, int bits_b, const unsigned char *b, double *out)
{
	int			rc = ora_check_bits(bits_a, bits_b);

	if (rc != ORA_OK)
		return rc;
	*out = (double) ora_bit_hamming((uint32_t) ((bits_a + 7) / 8), a, b);
	return ORA_OK;
}

/* jaccard_distance, src/bitvec.c:60-70 */
int
ora_jaccard_distance(int bits_a, const unsigned char *a, int bits_b, const unsigned char *b, double *out)
{
	int			rc = ora_check_bits(bits_a, bits_b);

	if (rc != ORA_OK)
		return rc;
	*out = ora_bit_jaccard((uint32_t) ((bits_a + 7) / 8), a, b);
	return ORA_OK;
}
