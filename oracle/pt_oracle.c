/*
 * pt_oracle.c -- CPU ORACLE (test infrastructure only; see pt_oracle.h header note: parity pinned by the reference's furnace images only, unpinned elsewhere).
 *
 * Plain-C restatement of the reference render loop.  Every function cites the reference
 * file:line it follows (paths relative to /root/reference/path_tracer/src unless stated).
 *
 * Arithmetic contract (shared, by construction, with the HIP kernel so that the two can be
 * compared bit-for-bit -- see DESIGN.md "deterministic math"):
 *   - IEEE binary32, round-to-nearest-even, no contraction (-ffp-contract=off); a fused
 *     multiply-add is used ONLY where written explicitly (fmaf): dot, cross, lerp, barycentric
 *     interpolation and the dm_* polynomials.
 *   - No libm transcendental is called.  sin/cos/tan/atan/atan2/asin/log/exp/pow are the dm_*
 *     functions below (Cody-Waite reduction + minimax polynomials in the style of Cephes'
 *     single-precision routines), accurate to a few ulp -- the same order as the CUDA libdevice
 *     functions the reference was built with, whose results are not reproducible here anyway.
 *   - sqrt and division are the correctly rounded IEEE operations.
 */
#define _GNU_SOURCE
#include "pt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* deterministic math                                                                         */
/* ------------------------------------------------------------------------------------------ */

static inline float dm_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
static inline float dm_sqrt(float x) { return __builtin_sqrtf(x); }
static inline float dm_abs(float x) { return __builtin_fabsf(x); }
/* fminf/fmaxf semantics (a NaN operand is ignored), spelled out so that both targets agree */
static inline float dm_min(float a, float b) { return (b != b || a < b) ? a : b; }
static inline float dm_max(float a, float b) { return (b != b || a > b) ? a : b; }
static inline int dm_isinf(float x) { return dm_abs(x) == INFINITY; }
static inline int dm_isnan(float x) { return x != x; }

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

#define DM_PI 3.14159265358979323f      /* types.hpp:9 */
#define DM_TWO_PI 6.28318530717958648f  /* types.hpp:10 */
#define DM_PI_OVER_TWO 1.57079632679489661f
#define DM_PI_OVER_FOUR 0.78539816339744830f
#define DM_INV_PI 0.31830988618379067f

/* sin and cos of x for |x| well below 2^22*pi/2 (the path only uses |x| < ~8) */
static inline void dm_sincos(float x, float* s, float* c)
{
    float kf = (x * 0.63661975f + 12582912.0f) - 12582912.0f; /* nearest integer to x*2/pi */
    int k = (int)kf;
    float r = dm_fma(kf, -1.5703125f, x);
    r = dm_fma(kf, -0.0004837512969970703f, r);
    r = dm_fma(kf, -7.549790126404332e-08f, r);
    float z = r * r;
    float sp = dm_fma(dm_fma(dm_fma(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    float cp = dm_fma(z * z, dm_fma(dm_fma(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                      dm_fma(-0.5f, z, 1.0f));
    switch (k & 3) {
    case 0: *s = sp; *c = cp; break;
    case 1: *s = cp; *c = -sp; break;
    case 2: *s = -sp; *c = -cp; break;
    default: *s = -cp; *c = sp; break;
    }
}
static inline float dm_sin(float x) { float s, c; dm_sincos(x, &s, &c); return s; }
static inline float dm_cos(float x) { float s, c; dm_sincos(x, &s, &c); return c; }
static inline float dm_tan(float x) { float s, c; dm_sincos(x, &s, &c); return s / c; }

static inline float dm_atan(float x)
{
    float ax = dm_abs(x);
    float y0, t;
    if (ax > 2.414213562373095f) { y0 = DM_PI_OVER_TWO; t = -(1.0f / ax); }
    else if (ax > 0.4142135623730950f) { y0 = DM_PI_OVER_FOUR; t = (ax - 1.0f) / (ax + 1.0f); }
    else { y0 = 0.0f; t = ax; }
    float z = t * t;
    float q = dm_fma(dm_fma(dm_fma(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
    float y = y0 + dm_fma(q * z, t, t);
    return (x < 0.0f) ? -y : y;
}

static inline float dm_atan2(float y, float x)
{
    if (x == 0.0f) {
        if (y == 0.0f) return 0.0f;
        return (y > 0.0f) ? DM_PI_OVER_TWO : -DM_PI_OVER_TWO;
    }
    float a = dm_atan(y / x);
    if (x < 0.0f) a = (y < 0.0f) ? a - DM_PI : a + DM_PI;
    return a;
}

static inline float dm_asin(float x)
{
    float a = dm_abs(x);
    if (a > 1.0f) return NAN;
    float z, w;
    int big = a > 0.5f;
    if (big) { z = 0.5f * (1.0f - a); w = dm_sqrt(z); }
    else { w = a; z = w * w; }
    float p = dm_fma(dm_fma(dm_fma(dm_fma(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z, 1.6666752422e-1f);
    float r = dm_fma(p * z, w, w);
    if (big) r = DM_PI_OVER_TWO - (r + r);
    return (x < 0.0f) ? -r : r;
}

/* natural log of a positive normal float */
static inline float dm_log(float x)
{
    if (!(x > 0.0f)) return (x == 0.0f) ? -INFINITY : NAN;
    if (x == INFINITY) return INFINITY;
    uint32_t bits = f2u(x);
    int e = (int)((bits >> 23) & 0xffu) - 126;
    float m = u2f((bits & 0x807fffffu) | 0x3f000000u); /* [0.5,1) */
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; }
    else { m = m - 1.0f; }
    float z = m * m;
    float p = dm_fma(7.0376836292e-2f, m, -1.1514610310e-1f);
    p = dm_fma(p, m, 1.1676998740e-1f);
    p = dm_fma(p, m, -1.2420140846e-1f);
    p = dm_fma(p, m, 1.4249322787e-1f);
    p = dm_fma(p, m, -1.6668057665e-1f);
    p = dm_fma(p, m, 2.0000714765e-1f);
    p = dm_fma(p, m, -2.4999993993e-1f);
    p = dm_fma(p, m, 3.3333331174e-1f);
    float y = (p * m) * z;
    float fe = (float)e;
    y = dm_fma(-2.12194440e-4f, fe, y);
    y = dm_fma(-0.5f, z, y);
    float r = m + y;
    r = dm_fma(0.693359375f, fe, r);
    return r;
}

static inline float dm_exp(float x)
{
    if (x != x) return x;
    if (x > 88.72283905206835f) return INFINITY;
    if (x < -103.278929903431851103f) return 0.0f;
    float nf = (x * 1.44269504088896341f + 12582912.0f) - 12582912.0f;
    int n = (int)nf;
    float r = dm_fma(nf, -0.693359375f, x);
    r = dm_fma(nf, 2.12194440e-4f, r);
    float z = r * r;
    float p = dm_fma(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = dm_fma(p, r, 8.3334519073e-3f);
    p = dm_fma(p, r, 4.1665795894e-2f);
    p = dm_fma(p, r, 1.6666665459e-1f);
    p = dm_fma(p, r, 5.0000001201e-1f);
    float y = dm_fma(p, z, r) + 1.0f;
    int n1 = n / 2, n2 = n - n1;
    float s1 = u2f((uint32_t)(n1 + 127) << 23);
    float s2 = u2f((uint32_t)(n2 + 127) << 23);
    return (y * s1) * s2;
}

/* ln(x) as an unevaluated sum hi + lo (x positive, normal, finite) */
static inline void dm_log_ext(float x, float* hi, float* lo)
{
    uint32_t bits = f2u(x);
    int e = (int)((bits >> 23) & 0xffu) - 126;
    float m = u2f((bits & 0x807fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; }
    else { m = m - 1.0f; }
    float z = m * m;
    float p = dm_fma(7.0376836292e-2f, m, -1.1514610310e-1f);
    p = dm_fma(p, m, 1.1676998740e-1f);
    p = dm_fma(p, m, -1.2420140846e-1f);
    p = dm_fma(p, m, 1.4249322787e-1f);
    p = dm_fma(p, m, -1.6668057665e-1f);
    p = dm_fma(p, m, 2.0000714765e-1f);
    p = dm_fma(p, m, -2.4999993993e-1f);
    p = dm_fma(p, m, 3.3333331174e-1f);
    float fe = (float)e;
    float c = (p * m) * z;
    c = dm_fma(-2.12194440e-4f, fe, c);
    /* -z/2 carried exactly: z = m*m rounded, zl = its rounding error */
    float zl = dm_fma(m, m, -z);
    c = dm_fma(-0.5f, zl, c);
    float a = 0.693359375f * fe; /* exact */
    /* a + m - z/2 in double-float */
    float s = a + m;
    float bb = s - a;
    float s_lo = (a - (s - bb)) + (m - bb);
    float hz = -0.5f * z;
    float s2 = s + hz;
    float bb2 = s2 - s;
    float s2_lo = (s - (s2 - bb2)) + (hz - bb2);
    float t = (s_lo + s2_lo) + c;
    float h = s2 + t;
    *hi = h;
    *lo = (s2 - h) + t;
}

/* x >= 0 only (the path never raises a negative base) */
static inline float dm_pow(float x, float y)
{
    if (y == 0.0f) return 1.0f;
    if (x == 1.0f) return 1.0f;
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : INFINITY;
    if (!(x > 0.0f) || x == INFINITY || x < 1.17549435e-38f) return dm_exp(y * dm_log(x));
    float lh, ll;
    dm_log_ext(x, &lh, &ll);
    float ph = y * lh;
    float pl = dm_fma(y, lh, -ph) + y * ll;
    float eh = dm_exp(ph);
    return dm_fma(eh, pl, eh);
}

float orc_dm_sin(float x) { return dm_sin(x); }
float orc_dm_cos(float x) { return dm_cos(x); }
float orc_dm_tan(float x) { return dm_tan(x); }
float orc_dm_atan(float x) { return dm_atan(x); }
float orc_dm_atan2(float y, float x) { return dm_atan2(y, x); }
float orc_dm_asin(float x) { return dm_asin(x); }
float orc_dm_log(float x) { return dm_log(x); }
float orc_dm_exp(float x) { return dm_exp(x); }
float orc_dm_pow(float x, float y) { return dm_pow(x, y); }

void orc_dm_batch(int fn, const float* x, const float* y, float* out, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        float a = x[i], b = y ? y[i] : 0.0f, r;
        switch (fn) {
        case 0: r = dm_sin(a); break;
        case 1: r = dm_cos(a); break;
        case 2: r = dm_tan(a); break;
        case 3: r = dm_atan(a); break;
        case 4: r = dm_atan2(a, b); break;
        case 5: r = dm_asin(a); break;
        case 6: r = dm_log(a); break;
        case 7: r = dm_exp(a); break;
        case 8: r = dm_pow(a, b); break;
        case 9: r = dm_sqrt(a); break;
        case 10: r = a / b; break;
        default: r = 0.0f;
        }
        out[i] = r;
    }
}

#ifdef ORC_HOST_LIBM
/* TOLERANCE-CALIBRATION BUILD ONLY (oracle/Makefile target libpt_oracle_hostlibm.so, tests/test_tolerance_calibration.py):
 * the same render loop with the host's libm transcendentals instead of the dm_* routines, compiled with -ffp-contract=fast.
 * BASELINE.md section 4 asks for exactly this pair - a second CPU build that differs the way an independent toolchain
 * (CUDA libdevice, nvcc's own fma choices) would - as the noise floor for the stated OptiX tolerance.  The orc_dm_* entry
 * points above keep exporting the deterministic routines. */
#define dm_sincos(x, s, c) sincosf((x), (s), (c))
#define dm_sin(x) sinf(x)
#define dm_cos(x) cosf(x)
#define dm_tan(x) tanf(x)
#define dm_atan(x) atanf(x)
#define dm_atan2(y, x) atan2f((y), (x))
#define dm_asin(x) asinf(x)
#define dm_log(x) logf(x)
#define dm_exp(x) expf(x)
#define dm_pow(x, y) powf((x), (y))
#endif

/* ------------------------------------------------------------------------------------------ */
/* vec3 (owl::vec3f stand-in; component-wise ops, see SURVEY 8(c) caveat on owl::normalize)    */
/* ------------------------------------------------------------------------------------------ */

typedef struct v3 { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vs(float s) { return V(s, s, s); }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return dm_fma(a.z, b.z, dm_fma(a.y, b.y, a.x * b.x)); }
static inline v3 vcross(v3 a, v3 b)
{
    return V(dm_fma(a.y, b.z, -(a.z * b.y)), dm_fma(a.z, b.x, -(a.x * b.z)), dm_fma(a.x, b.y, -(a.y * b.x)));
}
/* owl::normalize = v * rsqrt(dot(v,v)) with rsqrt(x) = 1/sqrt(x) */
static inline v3 vnormalize(v3 a) { return vscale(a, 1.0f / dm_sqrt(vdot(a, a))); }
static inline v3 ld3(const float* p) { return V(p[0], p[1], p[2]); }
static inline void st3(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

/* math.hpp:6-10 */
static inline float lerpf(float a, float b, float t) { return dm_fma(b - a, t, a); }
static inline v3 lerp3(v3 a, v3 b, float t) { return V(lerpf(a.x, b.x, t), lerpf(a.y, b.y, t), lerpf(a.z, b.z, t)); }
static inline float sqr(float v) { return v * v; } /* math.hpp:16 */

/* math.hpp:22-38 */
static inline float cos_theta(v3 w) { return w.z; }
static inline float sin_theta(v3 w) { return dm_sqrt(dm_max(0.0f, 1.0f - sqr(cos_theta(w)))); }
static inline float tan_theta(v3 w) { return sin_theta(w) / cos_theta(w); }
static inline float clampf(float x, float lo, float hi) { return dm_min(hi, dm_max(lo, x)); } /* disney_helper.cuh:14-17 */
static inline float cos_phi(v3 w)
{
    float theta = sin_theta(w);
    return (theta == 0.0f) ? 1.0f : clampf(w.x / theta, -1.0f, 1.0f);
}
static inline float sin_phi(v3 w)
{
    float theta = sin_theta(w);
    return (theta == 0.0f) ? 1.0f : clampf(w.y / theta, -1.0f, 1.0f); /* returns 1 at the pole: math.hpp:34-38 */
}

/* math.hpp:50-56 */
static inline v3 to_sphere3(float sin_t, float cos_t, float phi)
{
    float s, c;
    dm_sincos(phi, &s, &c);
    return V(sin_t * c, sin_t * s, cos_t);
}
/* math.hpp:42-48 */
static inline v3 to_sphere2(float theta, float phi)
{
    float st, ct;
    dm_sincos(theta, &st, &ct);
    return to_sphere3(st, ct, phi);
}

/* math.hpp:58-61 */
static inline v3 reflect(v3 w, v3 n) { return vsub(vscale(vscale(n, vdot(w, n)), 2.0f), w); }

/* math.hpp:63-77 */
static inline int refract(v3 w, v3 n, float eta, v3* wi)
{
    if (eta == 1.0f) { *wi = vneg(w); return 1; }
    float cos_theta_i = vdot(w, n);
    float sin2_theta_i = dm_max(0.0f, 1.0f - sqr(cos_theta_i));
    float sin2_theta_t = eta * eta * sin2_theta_i;
    if (sin2_theta_t > 1.0f) return 0;
    float cos_theta_t = dm_sqrt(1.0f - sin2_theta_t);
    *wi = vadd(vscale(vneg(w), eta), vscale(n, eta * cos_theta_i - cos_theta_t));
    return 1;
}

/* math.hpp:79-82 */
static inline int same_hemisphere(v3 a, v3 b) { return a.z * b.z > 0.0f; }

/* math.hpp:86-95 */
static inline void onb(v3 n, v3* t, v3* b)
{
    if (n.x != n.y || n.x != n.z) *t = V(n.z - n.y, n.x - n.z, n.y - n.x);
    else *t = V(n.z - n.y, n.x + n.z, -n.y - n.x);
    *t = vnormalize(*t);
    *b = vcross(n, *t);
}
/* math.hpp:98-101 */
static inline v3 to_local(v3 t, v3 b, v3 n, v3 w) { return vnormalize(V(vdot(w, t), vdot(w, b), vdot(w, n))); }
/* math.hpp:104-107 */
static inline v3 to_world(v3 t, v3 b, v3 n, v3 w)
{
    v3 r = V(dm_fma(w.z, n.x, dm_fma(w.y, b.x, w.x * t.x)), dm_fma(w.z, n.y, dm_fma(w.y, b.y, w.x * t.y)),
             dm_fma(w.z, n.z, dm_fma(w.y, b.z, w.x * t.z)));
    return vnormalize(r);
}

/* ------------------------------------------------------------------------------------------ */
/* RNG  random.hpp:34-85                                                                      */
/* ------------------------------------------------------------------------------------------ */

uint32_t orc_rng_init(uint32_t seed_u, uint32_t seed_v) /* random.hpp:46-56 */
{
    uint32_t s = 0;
    for (int n = 0; n < 4; n++) {
        s += 0x9e3779b9u;
        seed_u += ((seed_v << 4) + 0xa341316cu) ^ (seed_v + s) ^ ((seed_v >> 5) + 0xc8013ea4u);
        seed_v += ((seed_u << 4) + 0xad90777du) ^ (seed_u + s) ^ ((seed_u >> 5) + 0x7e95761eu);
    }
    return seed_u;
}
static inline float rng_next(uint32_t* state) /* random.hpp:61-69 */
{
    *state = 16807u * (*state) + 1013904223u;
    return (float)(*state) * 0x1p-32f; /* ldexpf((float)state, -32): exact scaling; can return 1.0f */
}
float orc_rng_next(uint32_t* state) { return rng_next(state); }

/* ------------------------------------------------------------------------------------------ */
/* sample_methods.hpp                                                                         */
/* ------------------------------------------------------------------------------------------ */

/* sample_methods.hpp:19-41 */
static inline void sample_concentric_disk(float rx, float ry, float* ox, float* oy)
{
    float dx = 2.0f * rx - 1.0f;
    float dy = 2.0f * ry - 1.0f;
    if (dx == 0.0f && dy == 0.0f) { *ox = 0.0f; *oy = 0.0f; return; }
    float phi, r;
    if (dm_abs(dx) > dm_abs(dy)) { r = dx; phi = DM_PI_OVER_FOUR * (dy / dx); }
    else { r = dy; phi = DM_PI_OVER_TWO - DM_PI_OVER_FOUR * (dx / dy); }
    float s, c;
    dm_sincos(phi, &s, &c);
    *ox = r * c;
    *oy = r * s;
}
/* sample_methods.hpp:53-60 */
static inline v3 sample_cosine_hemisphere(float rx, float ry)
{
    float cx, cy;
    sample_concentric_disk(rx, ry, &cx, &cy);
    float ct = dm_sqrt(dm_max(0.0f, 1.0f - sqr(cx) - sqr(cy)));
    return V(cx, cy, ct);
}
/* sample_methods.hpp:62-65 */
static inline float pdf_cosine_hemisphere(v3 wi) { return dm_abs(cos_theta(wi)) * DM_INV_PI; }

/* ------------------------------------------------------------------------------------------ */
/* material + Disney BSDF                                                                     */
/* ------------------------------------------------------------------------------------------ */

typedef struct material { /* device_global.hpp:19-36 */
    v3 base_color;
    float subsurface, metallic, specular, specular_tint, roughness, anisotropic, sheen, sheen_tint, clearcoat,
        clearcoat_gloss, ior, specular_transmission, specular_transmission_roughness, emission;
} material;

static material material_default(void)
{
    material m = {{0.8f, 0.8f, 0.8f}, 0.0f, 0.0f, 0.5f, 1.0f, 0.5f, 0.0f, 0.0f, 1.0f, 0.0f, 0.03f, 1.45f, 0.0f, 0.0f, 0.0f};
    return m;
}
static material material_load(const float* p)
{
    material m;
    m.base_color = V(p[0], p[1], p[2]);
    m.subsurface = p[3]; m.metallic = p[4]; m.specular = p[5]; m.specular_tint = p[6]; m.roughness = p[7];
    m.anisotropic = p[8]; m.sheen = p[9]; m.sheen_tint = p[10]; m.clearcoat = p[11]; m.clearcoat_gloss = p[12];
    m.ior = p[13]; m.specular_transmission = p[14]; m.specular_transmission_roughness = p[15]; m.emission = p[16];
    return m;
}

#define ALPHA_MIN 0.001f /* types.hpp:18 */

/* disney_helper.cuh:4-12 */
static inline v3 rgb_to_lin(v3 c) { return V(dm_pow(c.x, 2.2f), dm_pow(c.y, 2.2f), dm_pow(c.z, 2.2f)); }
static inline float luminance(v3 c) { return vdot(V(0.2126f, 0.7152f, 0.0722f), c); }
/* disney_helper.cuh:19-24 */
static inline float schlick_weight(float ct)
{
    float m = clampf(1.0f - ct, 0.0f, 1.0f);
    float m2 = m * m;
    return m2 * m2 * m;
}
/* disney_helper.cuh:31-37 */
static inline float relative_eta(v3 wo, float ior, float* eta_i, float* eta_t)
{
    *eta_i = cos_theta(wo) > 0.0f ? 1.0f : ior;
    *eta_t = cos_theta(wo) > 0.0f ? ior : 1.0f;
    return *eta_i / *eta_t;
}
/* disney_helper.cuh:39-42 */
static inline float roughness_to_alpha1(float roughness) { return dm_max(ALPHA_MIN, clampf(sqr(roughness), 0.0f, 1.0f)); }
/* disney_helper.cuh:44-48 */
static inline void roughness_to_alpha2(float roughness, float anisotropy, float* ax, float* ay)
{
    float aspect = dm_sqrt(1.0f - 0.9f * anisotropy);
    *ax = dm_max(ALPHA_MIN, sqr(roughness) / aspect);
    *ay = dm_max(ALPHA_MIN, sqr(roughness) * aspect);
}
/* disney_helper.cuh:52-60 */
static inline float fresnel_equation(v3 i, v3 m, float eta_i, float eta_t)
{
    float c = dm_abs(vdot(i, m));
    float denominator = sqr(eta_t / eta_i) - 1.0f + sqr(c);
    if (denominator < 0.0f) return 1.0f;
    float g = dm_sqrt(denominator);
    return 0.5f * sqr((g - c) / (g + c)) * (1.0f + sqr(c * (g + c) - 1.0f) / sqr(c * (g - c) + 1.0f));
}

/* disney_specular.cuh:17-27 */
static inline float lambda(v3 w, float ax, float ay)
{
    float abs_tan_theta = tan_theta(w);
    if (dm_isinf(abs_tan_theta)) return 0.0f;
    float alpha0 = dm_sqrt(sqr(cos_phi(w) * ax) + sqr(sin_phi(w) * ay));
    float a = 1.0f / (alpha0 * abs_tan_theta);
    return (-1.0f + dm_sqrt(1.0f + 1.0f / sqr(a))) / 2.0f;
}
/* disney_specular.cuh:31-34 */
static inline float g1_smith(v3 w, float ax, float ay) { return 1.0f / (1.0f + lambda(w, ax, ay)); }
/* disney_specular.cuh:38-41 */
static inline float g2_smith_separable(v3 wo, v3 wi, float ax, float ay) { return g1_smith(wo, ax, ay) * g1_smith(wi, ax, ay); }
/* disney_specular.cuh:46-49 */
static inline float g2_smith_correlated(v3 wo, v3 wi, float ax, float ay)
{
    return 1.0f / (1.0f + lambda(wo, ax, ay) + lambda(wi, ax, ay));
}
/* disney_specular.cuh:54-60 */
static inline float d_gtr_2(v3 wm, float ax, float ay)
{
    float tan2_theta = sqr(tan_theta(wm));
    if (dm_isinf(tan2_theta)) return 0.0f;
    float cos4_theta = sqr(sqr(cos_theta(wm)));
    float e = 1.0f + tan2_theta * (sqr(cos_phi(wm)) / sqr(ax) + sqr(sin_phi(wm)) / sqr(ay));
    return 1.0f / (DM_PI * ax * ay * cos4_theta * sqr(e));
}
/* disney_specular.cuh:64-81 (wo unused; "+ inv_pi" phase kept bug-for-bug, :69) */
static inline v3 sample_gtr2_ndf(float ax, float ay, float u0, float u1)
{
    float phi = dm_atan(ay / ax * dm_tan(DM_TWO_PI * u1 + DM_INV_PI));
    if (u1 > 0.5f) phi += DM_PI;
    float sin_p, cos_p;
    dm_sincos(phi, &sin_p, &cos_p);
    float alphax2 = sqr(ax), alphay2 = sqr(ay);
    float alpha2 = 1.0f / (sqr(cos_p) / alphax2 + sqr(sin_p) / alphay2);
    float tan_theta2 = alpha2 * u0 / (1.0f - u0);
    float cos_t = 1.0f / dm_sqrt(1.0f + tan_theta2);
    float sin_t = dm_sqrt(dm_max(0.0f, 1.0f - sqr(cos_t)));
    v3 wh = V(sin_t * cos_p, sin_t * sin_p, cos_t); /* to_sphere_coordinates(sin,cos,phi), math.hpp:50-56 */
    return vnormalize(wh);
}

/* disney_specular.cuh:125-149 */
static inline v3 eval_disney_specular_brdf(const material* m, v3 wo, v3 wh, v3 wi, float* pdf)
{
    float lum = luminance(m->base_color);
    v3 c_tint = lum > 0.0f ? vdivs(m->base_color, lum) : vs(1.0f);
    v3 c_spec = lerp3(vscale(lerp3(vs(1.0f), c_tint, m->specular_tint), 0.08f * m->specular), m->base_color, m->metallic);
    float ax, ay;
    roughness_to_alpha2(m->roughness, m->anisotropic, &ax, &ay);
    float d = d_gtr_2(wh, ax, ay);
    float g = g2_smith_correlated(wo, wi, ax, ay);
    v3 f = lerp3(c_spec, vs(1.0f), schlick_weight(vdot(wi, wh)));
    *pdf = d * g1_smith(wo, ax, ay) * dm_max(0.0f, vdot(wo, wh)) / (4.0f * cos_theta(wo));
    return vdivs(vscale(f, d * g), 4.0f * dm_abs(cos_theta(wo)));
}
/* disney_specular.cuh:151-170 */
static inline v3 sample_disney_specular_brdf(const material* m, v3 wo, uint32_t* rng, v3* wi, float* pdf)
{
    float ax, ay;
    roughness_to_alpha2(m->roughness, m->anisotropic, &ax, &ay);
    float u0 = rng_next(rng), u1 = rng_next(rng);
    v3 wh = sample_gtr2_ndf(ax, ay, u0, u1);
    if (vdot(wo, wh) < 0.0f) wh = vneg(wh);
    *wi = reflect(wo, wh);
    if (cos_theta(*wi) <= 0.0f) { *pdf = 0.0f; return vs(0.0f); }
    return eval_disney_specular_brdf(m, wo, wh, *wi, pdf);
}

/* disney_specular.cuh:175-180 */
static inline v3 sample_gtr2_bsdf(float a, float u0, float u1)
{
    float theta = dm_atan((a * dm_sqrt(u0)) / dm_sqrt(1.0f - u0));
    float phi = DM_TWO_PI * u1;
    return to_sphere2(theta, phi);
}
/* disney_specular.cuh:193-214 */
static inline v3 eval_disney_specular_bsdf(const material* m, v3 wo, v3 wh, v3 wi, float* pdf)
{
    float eta_i, eta_t;
    float eta = relative_eta(wo, m->ior, &eta_i, &eta_t);
    float R = fresnel_equation(wo, wh, eta_i, eta_t);
    float T = 1.0f - R;
    float pr = R, pt = T;
    if (same_hemisphere(wo, wi)) {
        *pdf = pr / (pr + pt);
        return vdivs(vscale(m->base_color, R), dm_abs(cos_theta(wi)));
    }
    *pdf = pt / (pr + pt);
    v3 sq = V(dm_sqrt(m->base_color.x), dm_sqrt(m->base_color.y), dm_sqrt(m->base_color.z));
    return vdivs(vdivs(vscale(sq, T), dm_abs(cos_theta(wi))), sqr(eta));
}
/* disney_specular.cuh:216-244 */
static inline v3 sample_disney_specular_bsdf(const material* m, v3 wo, uint32_t* rng, v3* wi, float* pdf)
{
    float u0 = rng_next(rng), u1 = rng_next(rng);
    v3 wh = sample_gtr2_bsdf(roughness_to_alpha1(m->specular_transmission_roughness), u0, u1);
    if (cos_theta(wo) < 0.0f && !same_hemisphere(wo, wh)) wh = vneg(wh);
    float eta_i, eta_t;
    float eta = relative_eta(wo, m->ior, &eta_i, &eta_t);
    float R = fresnel_equation(wo, wh, eta_i, eta_t);
    float T = 1.0f - R;
    float pr = R, pt = T;
    /* short-circuit ||: the third draw happens only if refraction succeeded (:235) */
    if (!refract(wo, wh, eta, wi) || rng_next(rng) < pr / (pr + pt)) {
        float ax, ay;
        roughness_to_alpha2(m->roughness, m->anisotropic, &ax, &ay);
        float v0 = rng_next(rng), v1 = rng_next(rng);
        wh = sample_gtr2_ndf(ax, ay, v0, v1);
        *wi = vnormalize(reflect(wo, wh));
    }
    return eval_disney_specular_bsdf(m, wo, wh, *wi, pdf);
}

/* disney_clearcoat.cuh:13-20 */
static inline float d_gtr1(v3 wh, float alpha)
{
    if (alpha >= 1.0f) return DM_INV_PI;
    float a2 = sqr(alpha);
    return (a2 - 1.0f) / (DM_PI * dm_log(a2) * (1.0f + (a2 - 1.0f) * sqr(cos_theta(wh))));
}
/* disney_clearcoat.cuh:23-33 */
static inline v3 sample_gtr1_ndf(v3 wo, float a, float u0, float u1)
{
    float alpha2 = sqr(a);
    float cos_t = dm_sqrt(dm_max(0.0f, (1.0f - dm_pow(alpha2, 1.0f - u0)) / (1.0f - alpha2)));
    float sin_t = dm_sqrt(dm_max(0.0f, 1.0f - sqr(cos_t)));
    float phi = DM_TWO_PI * u1;
    v3 wh = to_sphere3(sin_t, cos_t, phi);
    if (!same_hemisphere(wo, wh)) wh = vneg(wh);
    return wh;
}
/* disney_clearcoat.cuh:45-59 (Fresnel lerp argument order kept bug-for-bug, :54) */
static inline v3 eval_disney_clearcoat(const material* m, v3 wo, v3 wh, v3 wi, float* pdf)
{
    if (m->clearcoat <= 0.0f) { *pdf = 0.0f; return vs(0.0f); }
    float d = d_gtr1(wh, lerpf(0.1f, 0.001f, m->clearcoat_gloss));
    float f = lerpf(1.0f, schlick_weight(cos_theta(wi)), 0.04f);
    float g = g2_smith_separable(wo, wi, 0.25f, 0.25f);
    *pdf = d / (4.0f * vdot(wh, wi));
    return vs(d * g * f / (4.0f * dm_abs(cos_theta(wo)) * dm_abs(cos_theta(wi))));
}
/* disney_clearcoat.cuh:61-78 */
static inline v3 sample_disney_clearcoat(const material* m, v3 wo, uint32_t* rng, v3* wi, float* pdf)
{
    float a = lerpf(0.1f, 0.001f, m->clearcoat_gloss);
    float u0 = rng_next(rng), u1 = rng_next(rng);
    v3 wh = sample_gtr1_ndf(wo, a, u0, u1);
    if (vdot(wh, wo) < 0.0f) wh = vneg(wh);
    wh = vnormalize(wh);
    *wi = reflect(wo, wh);
    if (!same_hemisphere(wo, *wi)) { *pdf = 0.0f; return vs(0.0f); }
    return eval_disney_clearcoat(m, wo, wh, *wi, pdf);
}

/* disney_diffuse.cuh:26-55 (subsurface unused) */
static inline v3 eval_disney_diffuse(const material* m, v3 wo, v3 wi, float* pdf)
{
    float cos_theta_o = cos_theta(wo);
    float cos_theta_i = cos_theta(wi);
    float fresnel_o = schlick_weight(cos_theta_o);
    float fresnel_i = schlick_weight(cos_theta_i);
    v3 lambert = vscale(m->base_color, DM_INV_PI);
    float fd = (1.0f - 0.5f * fresnel_o) * (1.0f - 0.5f * fresnel_i);
    float rr = m->roughness * (vdot(wo, wi) + 1.0f);
    float fr = rr * (fresnel_i + fresnel_o + fresnel_o * fresnel_i * (rr - 1.0f));
    *pdf = pdf_cosine_hemisphere(wi);
    return vscale(lambert, fd + fr);
}
/* disney_diffuse.cuh:57-62 */
static inline v3 sample_disney_diffuse(const material* m, v3 wo, uint32_t* rng, v3* wi, float* pdf)
{
    float u0 = rng_next(rng), u1 = rng_next(rng);
    *wi = sample_cosine_hemisphere(u0, u1);
    return eval_disney_diffuse(m, wo, *wi, pdf);
}

/* disney_sheen.cuh:15-37 */
static inline v3 eval_disney_sheen(const material* m, v3 wo, v3 wi)
{
    if (m->sheen <= 0.0f) return vs(0.0f);
    v3 wh = vadd(wi, wo);
    if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return vs(0.0f);
    wh = vnormalize(wh);
    float lum = luminance(rgb_to_lin(m->base_color));
    float cos_theta_d = vdot(wi, wh);
    v3 tint = (lum > 0.0f) ? vdivs(m->base_color, lum) : vs(1.0f);
    return vscale(vscale(lerp3(vs(1.0f), tint, m->sheen_tint), m->sheen), schlick_weight(cos_theta_d));
}

#define LOBE_NONE (-1)
#define LOBE_DIFFUSE 0
#define LOBE_CLEARCOAT 1
#define LOBE_METALLIC 2
#define LOBE_GLASS 3

/* disney.cuh:15-29 */
static inline void calculate_pdf_of_lobes(const material* m, float* p_metallic, float* p_diffuse, float* p_clearcoat, float* p_glass)
{
    float diffuse_weight = (1.0f - m->specular_transmission) * (1.0f - m->metallic);
    float metallic_weight = m->metallic;
    float clearcoat_weight = 0.25f * m->clearcoat;
    float glass_weight = (1.0f - m->metallic) * m->specular_transmission;
    float factor = 1.0f / (metallic_weight + glass_weight + diffuse_weight + clearcoat_weight);
    *p_metallic = metallic_weight * factor;
    *p_glass = glass_weight * factor;
    *p_diffuse = diffuse_weight * factor;
    *p_clearcoat = clearcoat_weight * factor;
}

/* disney.cuh:31-66 */
static inline v3 sample_disney(const material* m, v3 wo, uint32_t* rng, v3* wi, float* pdf, int* sampled_lobe)
{
    float p_metallic, p_diffuse, p_clearcoat, p_glass;
    calculate_pdf_of_lobes(m, &p_metallic, &p_diffuse, &p_clearcoat, &p_glass);
    int force_btdf = cos_theta(wo) < 0.0f && *sampled_lobe == LOBE_GLASS;
    float p = rng_next(rng);
    v3 f = vs(0.0f);
    if (!force_btdf && p <= p_metallic) {
        f = sample_disney_specular_brdf(m, wo, rng, wi, pdf);
        *sampled_lobe = LOBE_METALLIC;
    } else if (!force_btdf && p > p_metallic && p <= (p_metallic + p_clearcoat)) {
        f = sample_disney_clearcoat(m, wo, rng, wi, pdf);
        *sampled_lobe = LOBE_CLEARCOAT;
    } else if (!force_btdf && p > p_metallic + p_clearcoat && p <= (p_metallic + p_clearcoat + p_diffuse)) {
        f = sample_disney_diffuse(m, wo, rng, wi, pdf);
        *sampled_lobe = LOBE_DIFFUSE;
    } else if (force_btdf || p_glass >= 0.0f) {
        f = sample_disney_specular_bsdf(m, wo, rng, wi, pdf);
        *sampled_lobe = LOBE_GLASS;
    }
    return vadd(f, eval_disney_sheen(m, wo, *wi));
}

/* ------------------------------------------------------------------------------------------ */
/* unit hooks                                                                                 */
/* ------------------------------------------------------------------------------------------ */

void orc_sample_disney(const float mat[ORC_MAT_FLOATS], const float wo[3], uint32_t* rng_state, int32_t* sampled_lobe,
                       float f[3], float wi[3], float* pdf)
{
    material m = material_load(mat);
    v3 lwi = vs(0.0f); /* device.cu:181 local_wi{} */
    float lpdf = 0.0f; /* device.cu:183 */
    int lobe = *sampled_lobe;
    v3 r = sample_disney(&m, ld3(wo), rng_state, &lwi, &lpdf, &lobe);
    *sampled_lobe = lobe;
    st3(f, r);
    st3(wi, lwi);
    *pdf = lpdf;
}
void orc_onb(const float n[3], float t[3], float b[3]) { v3 tt, bb; onb(ld3(n), &tt, &bb); st3(t, tt); st3(b, bb); }
void orc_to_local(const float t[3], const float b[3], const float n[3], const float w[3], float out[3])
{
    st3(out, to_local(ld3(t), ld3(b), ld3(n), ld3(w)));
}
void orc_to_world(const float t[3], const float b[3], const float n[3], const float w[3], float out[3])
{
    st3(out, to_world(ld3(t), ld3(b), ld3(n), ld3(w)));
}
void orc_sample_cosine_hemisphere(float u0, float u1, float out[3]) { st3(out, sample_cosine_hemisphere(u0, u1)); }
int orc_refract(const float w[3], const float n[3], float eta, float wi[3])
{
    v3 r = vs(0.0f);
    int ok = refract(ld3(w), ld3(n), eta, &r);
    st3(wi, r);
    return ok;
}
float orc_fresnel_equation(const float i[3], const float m[3], float eta_i, float eta_t) { return fresnel_equation(ld3(i), ld3(m), eta_i, eta_t); }
float orc_d_gtr1(const float wh[3], float alpha) { return d_gtr1(ld3(wh), alpha); }
float orc_d_gtr2(const float wm[3], float ax, float ay) { return d_gtr_2(ld3(wm), ax, ay); }
float orc_lambda(const float w[3], float ax, float ay) { return lambda(ld3(w), ax, ay); }
void orc_eval_lobe(int lobe, const float mat[ORC_MAT_FLOATS], const float wo[3], const float wh[3], const float wi[3], float f[3], float* pdf)
{
    material m = material_load(mat);
    v3 r = vs(0.0f);
    float p = 0.0f;
    switch (lobe) {
    case LOBE_DIFFUSE: r = eval_disney_diffuse(&m, ld3(wo), ld3(wi), &p); break;
    case LOBE_CLEARCOAT: r = eval_disney_clearcoat(&m, ld3(wo), ld3(wh), ld3(wi), &p); break;
    case LOBE_METALLIC: r = eval_disney_specular_brdf(&m, ld3(wo), ld3(wh), ld3(wi), &p); break;
    case LOBE_GLASS: r = eval_disney_specular_bsdf(&m, ld3(wo), ld3(wh), ld3(wi), &p); break;
    default: break;
    }
    st3(f, r);
    *pdf = p;
}
void orc_eval_sheen(const float mat[ORC_MAT_FLOATS], const float wo[3], const float wi[3], float f[3])
{
    material m = material_load(mat);
    st3(f, eval_disney_sheen(&m, ld3(wo), ld3(wi)));
}

/* owl::make_rgba -- UNVERIFIED (OWL source absent): make_8bit(f)=min(255,max(0,int(f*256.f))) */
static inline uint32_t make_8bit(float f)
{
    float s = f * 256.0f;
    int v = (s != s) ? 0 : (s >= 2147483520.0f ? 2147483647 : (s <= -2147483520.0f ? -2147483647 : (int)s));
    if (v < 0) v = 0;
    if (v > 255) v = 255;
    return (uint32_t)v;
}
static inline uint32_t make_rgba(v3 c) { return make_8bit(c.x) | (make_8bit(c.y) << 8) | (make_8bit(c.z) << 16) | (0xffu << 24); }
uint32_t orc_make_rgba(const float c[3]) { return make_rgba(ld3(c)); }

/* camera.cpp:3-21 (host code in the reference: uses the host libm tan, as the reference does) */
void orc_to_camera_data(const float look_from[3], const float look_at[3], const float look_up[3], float vertical_fov, int w, int h,
                        orc_camera* out)
{
    float aspect = (float)w / (float)h;
    float theta = vertical_fov * DM_PI / 180.0f;
    float hh = tanf(theta / 2);
    float viewport_height = 2.0f * hh;
    float viewport_width = aspect * viewport_height;
    v3 origin = ld3(look_from);
    v3 ww = vnormalize(vsub(ld3(look_from), ld3(look_at)));
    v3 u = vnormalize(vcross(ld3(look_up), ww));
    v3 v = vnormalize(vcross(ww, u));
    v3 horizontal = vscale(u, viewport_width);
    v3 vertical = vscale(v, viewport_height);
    v3 llc = vsub(vsub(vsub(origin, vdivs(horizontal, 2.0f)), vdivs(vertical, 2.0f)), ww);
    st3(out->origin, origin);
    st3(out->llc, llc);
    st3(out->horizontal, horizontal);
    st3(out->vertical, vertical);
}

/* ------------------------------------------------------------------------------------------ */
/* textures: tex2D<float4>, RGBA8 normalised-float read, nearest, clamp, normalised coords     */
/* (owl.hpp:248-257, application.cpp:236-240)                                                  */
/* ------------------------------------------------------------------------------------------ */

static inline int tex_coord(float u, int n)
{
    float x = u * (float)n;
    float fl = __builtin_floorf(x);
    int i = (fl != fl) ? 0 : (fl >= (float)n ? n - 1 : (fl < 0.0f ? 0 : (int)fl));
    return i;
}
static inline v3 tex_nearest(const orc_texture* t, float u, float v)
{
    int ix = tex_coord(u, t->width), iy = tex_coord(v, t->height);
    uint32_t p = t->rgba8[(size_t)iy * (size_t)t->width + (size_t)ix];
    return V((float)(p & 0xffu) / 255.0f, (float)((p >> 8) & 0xffu) / 255.0f, (float)((p >> 16) & 0xffu) / 255.0f);
}
void orc_tex_nearest(const orc_texture* t, float u, float v, float rgb[3]) { st3(rgb, tex_nearest(t, u, v)); }

/* device.cu:23-28 */
static inline void uv_on_sphere(v3 n, float* u, float* v)
{
    *u = 0.5f + dm_atan2(n.x, n.z) / (2.0f * DM_PI);
    *v = 0.5f + dm_asin(n.y) / DM_PI;
}
void orc_uv_on_sphere(const float n[3], float uv[2]) { uv_on_sphere(ld3(n), &uv[0], &uv[1]); }

/* ------------------------------------------------------------------------------------------ */
/* scene + BVH (the reference's traversal is OptiX, source absent -- own design)    */
/* ------------------------------------------------------------------------------------------ */

typedef struct bnode {
    float lmin[3], lmax[3], rmin[3], rmax[3];
    int32_t left, right; /* >=0 internal node; <0: ~((first<<3)|count) */
} bnode;

struct orc_scene {
    int n_tris;
    float* pos;   /* n*9, BVH (leaf) order */
    int32_t* ids; /* leaf order -> global triangle id */
    float* nrm;   /* n*9 global order */
    float* tc;    /* n*6 global order or NULL */
    float* gpos;  /* n*9 global order */
    int32_t* mat_idx;
    int32_t* tex_idx;
    int n_materials;
    float* materials;
    int n_textures;
    orc_texture* textures;
    bnode* nodes;
    int n_nodes;
    int depth;
    int32_t root; /* root child ref (leaf if tiny) */
    float rmin[3], rmax[3];
};

typedef struct build_ctx {
    orc_scene* s;
    const float* gpos;
    float* cen; /* n*3 */
    int32_t* order;
    int leaf_size;
    float pad;
    int node_cap;
} build_ctx;

static void tri_bounds(const float* p, float mn[3], float mx[3])
{
    for (int a = 0; a < 3; ++a) {
        float v0 = p[a], v1 = p[3 + a], v2 = p[6 + a];
        float lo = v0 < v1 ? v0 : v1; lo = lo < v2 ? lo : v2;
        float hi = v0 > v1 ? v0 : v1; hi = hi > v2 ? hi : v2;
        if (lo < mn[a]) mn[a] = lo;
        if (hi > mx[a]) mx[a] = hi;
    }
}
static void range_bounds(build_ctx* c, int lo, int hi, float mn[3], float mx[3])
{
    for (int a = 0; a < 3; ++a) { mn[a] = INFINITY; mx[a] = -INFINITY; }
    for (int i = lo; i < hi; ++i) tri_bounds(c->gpos + (size_t)c->order[i] * 9, mn, mx);
    for (int a = 0; a < 3; ++a) { mn[a] -= c->pad; mx[a] += c->pad; }
}

static int g_axis;
static const float* g_cen;
static int cmp_centroid(const void* a, const void* b)
{
    int32_t ia = *(const int32_t*)a, ib = *(const int32_t*)b;
    float ca = g_cen[(size_t)ia * 3 + g_axis], cb = g_cen[(size_t)ib * 3 + g_axis];
    if (ca < cb) return -1;
    if (ca > cb) return 1;
    return (ia > ib) - (ia < ib);
}
/* nth_element-style partition (quickselect) on centroid[axis] with id tie-break */
static void select_median(build_ctx* c, int lo, int hi, int mid, int axis)
{
    int32_t* o = c->order;
    const float* cen = c->cen;
    while (hi - lo > 16) {
        int m = lo + (hi - lo) / 2;
        /* median of three pivot */
        int32_t a = o[lo], b = o[m], d = o[hi - 1];
        float ca = cen[(size_t)a * 3 + axis], cb = cen[(size_t)b * 3 + axis], cd = cen[(size_t)d * 3 + axis];
        float pv; int32_t pid;
        if ((ca <= cb && cb <= cd) || (cd <= cb && cb <= ca)) { pv = cb; pid = b; }
        else if ((cb <= ca && ca <= cd) || (cd <= ca && ca <= cb)) { pv = ca; pid = a; }
        else { pv = cd; pid = d; }
        int i = lo, j = hi - 1;
        while (i <= j) {
            for (;;) { float v = cen[(size_t)o[i] * 3 + axis]; if (v < pv || (v == pv && o[i] < pid)) ++i; else break; }
            for (;;) { float v = cen[(size_t)o[j] * 3 + axis]; if (v > pv || (v == pv && o[j] > pid)) --j; else break; }
            if (i <= j) { int32_t t = o[i]; o[i] = o[j]; o[j] = t; ++i; --j; }
        }
        if (mid <= j) hi = j + 1;
        else if (mid >= i) lo = i;
        else return;
    }
    g_axis = axis; g_cen = cen;
    qsort(o + lo, (size_t)(hi - lo), sizeof(int32_t), cmp_centroid);
}

static int32_t build_rec(build_ctx* c, int lo, int hi, int depth)
{
    orc_scene* s = c->s;
    if (depth > s->depth) s->depth = depth;
    int n = hi - lo;
    if (n <= c->leaf_size) return ~((lo << 3) | n);
    /* split at the object median along the largest centroid extent */
    float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = lo; i < hi; ++i)
        for (int a = 0; a < 3; ++a) {
            float v = c->cen[(size_t)c->order[i] * 3 + a];
            if (v < cmn[a]) cmn[a] = v;
            if (v > cmx[a]) cmx[a] = v;
        }
    int axis = 0;
    float ext = cmx[0] - cmn[0];
    if (cmx[1] - cmn[1] > ext) { axis = 1; ext = cmx[1] - cmn[1]; }
    if (cmx[2] - cmn[2] > ext) { axis = 2; }
    int mid = lo + n / 2;
    select_median(c, lo, hi, mid, axis);
    int idx = s->n_nodes++;
    bnode* nd = &s->nodes[idx];
    range_bounds(c, lo, mid, nd->lmin, nd->lmax);
    range_bounds(c, mid, hi, nd->rmin, nd->rmax);
    int32_t l = build_rec(c, lo, mid, depth + 1);
    int32_t r = build_rec(c, mid, hi, depth + 1);
    s->nodes[idx].left = l;
    s->nodes[idx].right = r;
    return idx;
}

static void* xmemdup(const void* p, size_t n)
{
    if (!p || !n) return NULL;
    void* r = malloc(n);
    memcpy(r, p, n);
    return r;
}

orc_scene* orc_scene_create(const orc_scene_desc* d, int leaf_size)
{
    orc_scene* s = (orc_scene*)calloc(1, sizeof(orc_scene));
    int n = d->n_tris;
    s->n_tris = n;
    s->gpos = (float*)xmemdup(d->positions, (size_t)n * 9 * 4);
    /* Closest-hit definition, part 2: slivers are never hit.  A triangle whose height over its longest edge is below 1e-5
       of that edge (4 A^2 <= 1e-10 L^4, evaluated in double) is replaced by a point: for such a needle Moeller-Trumbore's u, v, t are
       rounding noise, it can report a "hit" far outside the triangle's bounding box, and whether a BVH walk ever tests the triangle for
       that ray then depends on the visiting order - the minimum over all triangles would not be what any hierarchy computes. */
    for (int i = 0; i < n; ++i) {
        float* q = s->gpos + (size_t)i * 9;
        double a[3], b[3], cc[3], la = 0.0, lb = 0.0, lc = 0.0;
        for (int k = 0; k < 3; ++k) {
            a[k] = (double)q[3 + k] - (double)q[k];
            b[k] = (double)q[6 + k] - (double)q[k];
            cc[k] = (double)q[6 + k] - (double)q[3 + k];
        }
        for (int k = 0; k < 3; ++k) { la += a[k] * a[k]; lb += b[k] * b[k]; lc += cc[k] * cc[k]; }
        double nx = a[1] * b[2] - a[2] * b[1], ny = a[2] * b[0] - a[0] * b[2], nz = a[0] * b[1] - a[1] * b[0];
        double four_area2 = nx * nx + ny * ny + nz * nz;
        double longest2 = la > lb ? (la > lc ? la : lc) : (lb > lc ? lb : lc);
        if (!(four_area2 > 1e-10 * longest2 * longest2)) {
            for (int k = 0; k < 3; ++k) q[3 + k] = q[6 + k] = q[k];
        }
    }
    s->nrm = (float*)xmemdup(d->normals, (size_t)n * 9 * 4);
    s->tc = (float*)xmemdup(d->texcoords, (size_t)n * 6 * 4);
    s->mat_idx = (int32_t*)xmemdup(d->material_index, (size_t)n * 4);
    s->tex_idx = (int32_t*)xmemdup(d->texture_index, (size_t)n * 4);
    s->n_materials = d->n_materials;
    s->materials = (float*)xmemdup(d->materials, (size_t)d->n_materials * ORC_MAT_FLOATS * 4);
    s->n_textures = d->n_textures;
    if (d->n_textures > 0) {
        s->textures = (orc_texture*)calloc((size_t)d->n_textures, sizeof(orc_texture));
        for (int i = 0; i < d->n_textures; ++i) {
            s->textures[i].width = d->textures[i].width;
            s->textures[i].height = d->textures[i].height;
            s->textures[i].rgba8 = (const uint32_t*)xmemdup(d->textures[i].rgba8, (size_t)d->textures[i].width * d->textures[i].height * 4);
        }
    }
    if (leaf_size < 1) leaf_size = 4;
    if (leaf_size > 7) leaf_size = 7;
    build_ctx c;
    c.s = s;
    c.gpos = s->gpos;
    c.leaf_size = leaf_size;
    c.cen = (float*)malloc((size_t)(n > 0 ? n : 1) * 3 * 4);
    c.order = (int32_t*)malloc((size_t)(n > 0 ? n : 1) * 4);
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n; ++i) {
        const float* p = s->gpos + (size_t)i * 9;
        c.order[i] = i;
        for (int a = 0; a < 3; ++a) c.cen[(size_t)i * 3 + a] = (p[a] + p[3 + a] + p[6 + a]) * (1.0f / 3.0f);
        tri_bounds(p, mn, mx);
    }
    float ext = 0.0f;
    for (int a = 0; a < 3; ++a) {
        float e = n > 0 ? mx[a] - mn[a] : 0.0f;
        if (e > ext) ext = e;
        float m = n > 0 ? (dm_abs(mn[a]) > dm_abs(mx[a]) ? dm_abs(mn[a]) : dm_abs(mx[a])) : 0.0f;
        if (m > ext) ext = m;
    }
    /* boxes are padded so that the slab test is conservative with respect to every hit the
       Moeller-Trumbore test can report; closest-hit is then independent of BVH topology */
    c.pad = ext * 1e-5f;
    s->nodes = (bnode*)malloc(sizeof(bnode) * (size_t)(n > 0 ? n : 1));
    s->n_nodes = 0;
    s->depth = 0;
    if (n > 0) {
        s->root = build_rec(&c, 0, n, 1);
        range_bounds(&c, 0, n, s->rmin, s->rmax);
    } else {
        s->root = ~0;
    }
    s->pos = (float*)malloc((size_t)(n > 0 ? n : 1) * 9 * 4);
    s->ids = c.order;
    for (int i = 0; i < n; ++i) memcpy(s->pos + (size_t)i * 9, s->gpos + (size_t)c.order[i] * 9, 36);
    free(c.cen);
    return s;
}

void orc_scene_destroy(orc_scene* s)
{
    if (!s) return;
    free(s->pos); free(s->ids); free(s->nrm); free(s->tc); free(s->gpos); free(s->mat_idx); free(s->tex_idx);
    free(s->materials);
    for (int i = 0; i < s->n_textures; ++i) free((void*)s->textures[i].rgba8);
    free(s->textures);
    free(s->nodes);
    free(s);
}
void orc_scene_set_materials(orc_scene* s, const float* materials, int n_materials)
{
    free(s->materials);
    s->n_materials = n_materials;
    s->materials = (float*)xmemdup(materials, (size_t)n_materials * ORC_MAT_FLOATS * 4);
}
int orc_scene_bvh_nodes(const orc_scene* s) { return s->n_nodes; }
int orc_scene_bvh_depth(const orc_scene* s) { return s->depth; }

typedef struct hit { float t, u, v; int32_t prim; } hit;

/* Moeller-Trumbore, two-sided, tmin < t < tmax; ties in t resolved towards the lower global id */
static inline void tri_test(const float* p, int32_t id, v3 o, v3 d, float tmin, hit* h)
{
    v3 p0 = ld3(p), p1 = ld3(p + 3), p2 = ld3(p + 6);
    v3 e1 = vsub(p1, p0), e2 = vsub(p2, p0);
    v3 pv = vcross(d, e2);
    float det = vdot(e1, pv);
    float inv = 1.0f / det;
    v3 tv = vsub(o, p0);
    float u = vdot(tv, pv) * inv;
    v3 qv = vcross(tv, e1);
    float v = vdot(d, qv) * inv;
    float t = vdot(e2, qv) * inv;
    if (u >= 0.0f && v >= 0.0f && u + v <= 1.0f && t > tmin && (t < h->t || (t == h->t && id < h->prim))) {
        h->t = t; h->u = u; h->v = v; h->prim = id;
    }
}

static inline int box_test(const float* bmin, const float* bmax, v3 o, v3 inv, float tmin, float tbest, float* tnear)
{
    float t0x = (bmin[0] - o.x) * inv.x, t1x = (bmax[0] - o.x) * inv.x;
    float t0y = (bmin[1] - o.y) * inv.y, t1y = (bmax[1] - o.y) * inv.y;
    float t0z = (bmin[2] - o.z) * inv.z, t1z = (bmax[2] - o.z) * inv.z;
    float tn = dm_max(dm_max(dm_min(t0x, t1x), dm_min(t0y, t1y)), dm_max(dm_min(t0z, t1z), tmin));
    float tf = dm_min(dm_min(dm_max(t0x, t1x), dm_max(t0y, t1y)), dm_min(dm_max(t0z, t1z), tbest));
    *tnear = tn;
    return tn <= tf * 1.0000004f;
}

static int intersect_bvh(const orc_scene* s, v3 o, v3 d, float tmin, float tmax, hit* h, orc_counters* cnt)
{
    h->t = tmax; h->prim = 0x7fffffff; h->u = h->v = 0.0f;
    if (s->n_tris == 0) return 0;
    v3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int32_t stack[128];
    int sp = 0;
    int32_t cur = s->root;
    for (;;) {
        if (cur >= 0) {
            const bnode* nd = &s->nodes[cur];
            if (cnt) cnt->nodes++;
            float tl, tr;
            int hl = box_test(nd->lmin, nd->lmax, o, inv, tmin, h->t, &tl);
            int hr = box_test(nd->rmin, nd->rmax, o, inv, tmin, h->t, &tr);
            if (hl && hr) {
                int32_t nearc = nd->left, farc = nd->right;
                if (tr < tl) { nearc = nd->right; farc = nd->left; }
                stack[sp++] = farc;
                cur = nearc;
                continue;
            } else if (hl) { cur = nd->left; continue; }
            else if (hr) { cur = nd->right; continue; }
        } else {
            uint32_t code = (uint32_t)~cur;
            int first = (int)(code >> 3), count = (int)(code & 7u);
            for (int i = 0; i < count; ++i) {
                if (cnt) cnt->tris++;
                tri_test(s->pos + (size_t)(first + i) * 9, s->ids[first + i], o, d, tmin, h);
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    return h->prim != 0x7fffffff;
}

static int intersect_brute(const orc_scene* s, v3 o, v3 d, float tmin, float tmax, hit* h, orc_counters* cnt)
{
    h->t = tmax; h->prim = 0x7fffffff; h->u = h->v = 0.0f;
    for (int i = 0; i < s->n_tris; ++i) {
        if (cnt) cnt->tris++;
        tri_test(s->gpos + (size_t)i * 9, i, o, d, tmin, h);
    }
    return h->prim != 0x7fffffff;
}

int orc_intersect(const orc_scene* s, const float org[3], const float dir[3], float tmin, float tmax, int use_bvh, float* t, float* u,
                  float* v, int32_t* prim)
{
    hit h;
    int ok = use_bvh ? intersect_bvh(s, ld3(org), ld3(dir), tmin, tmax, &h, NULL) : intersect_brute(s, ld3(org), ld3(dir), tmin, tmax, &h, NULL);
    *t = h.t; *u = h.u; *v = h.v; *prim = ok ? h.prim : -1;
    return ok;
}

/* ------------------------------------------------------------------------------------------ */
/* trace_path  device.cu:113-218                                                              */
/* ------------------------------------------------------------------------------------------ */

#define T_MIN 1e-3f /* types.hpp:16 */
#define T_MAX 1e10f /* types.hpp:17 */

typedef struct render_ctx {
    const orc_scene* s;
    const orc_env* env;
    int max_depth;
    int use_bvh;
} render_ctx;

/* Per-bounce log of ONE sample (orc_trace_sample): 32 floats per bounce - org 0-2, dir 3-5, hit flag 6, t u v 7-9, prim 10 (int bits),
   material index 11 (int bits), rng state before sample_disney 12 (uint bits), local_wo 13-15, f 16-18, pdf 19, local_wi 20-22,
   sampled lobe 23 (int bits), throughput after the bounce 24-26, rng state after the bounce 27 (uint bits), radiance 28-30, depth 31. */
#define ORC_LOG_FLOATS 32
static __thread float* g_log = NULL;
static __thread int g_log_rows = 0, g_log_cap = 0;
static float* log_row(void)
{
    if (!g_log || g_log_rows >= g_log_cap) return NULL;
    float* r = g_log + (size_t)g_log_rows++ * ORC_LOG_FLOATS;
    memset(r, 0, ORC_LOG_FLOATS * sizeof(float));
    return r;
}
static float bits_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static v3 trace_path(const render_ctx* rc, v3 org, v3 dir, uint32_t* rng, orc_counters* cnt)
{
    const orc_scene* s = rc->s;
    const orc_env* env = rc->env;
    v3 radiance = vs(0.0f);
    v3 throughput = vs(1.0f);
    int sampled_lobe = LOBE_NONE;
    int retries = 0;

    for (int depth = 0; depth < rc->max_depth; ++depth) {
        hit h;
        /* owl::traceRay, device.cu:133 (closest hit over all triangles, two-sided) */
        int got = rc->use_bvh ? intersect_bvh(s, org, dir, T_MIN, T_MAX, &h, cnt) : intersect_brute(s, org, dir, T_MIN, T_MAX, &h, cnt);
        if (cnt) cnt->rays++;
        float* lg = log_row();
        if (lg) {
            st3(lg, org); st3(lg + 3, dir); lg[6] = (float)got; lg[31] = (float)depth;
            if (got) { lg[7] = h.t; lg[8] = h.u; lg[9] = h.v; lg[10] = bits_f((uint32_t)h.prim); }
        }

        if (!got) { /* device.cu:136-148 */
            if (env->use_map && env->map.width > 0) {
                float tu, tv;
                uv_on_sphere(dir, &tu, &tv);
                radiance = vadd(radiance, tex_nearest(&env->map, tu, tv));
                if (cnt) cnt->env_misses++;
            } else if (env->use_auto) {
                radiance = vadd(radiance, lerp3(vs(1.0f), V(0.5f, 0.7f, 1.0f), 0.5f * (dir.y + 1.0f)));
            } else {
                radiance = vadd(radiance, ld3(env->color));
            }
            radiance = vscale(radiance, env->intensity);
            break;
        }

        /* triangle_hit, device.cu:256-287 */
        v3 wo = vneg(vnormalize(dir));
        int32_t prim = h.prim;
        int32_t mi = s->mat_idx[prim];

        material mat = material_default(); /* device.cu:150-154 */
        if (mi >= 0) mat = material_load(s->materials + (size_t)mi * ORC_MAT_FLOATS);

        if (mat.emission > 0.0f) { /* device.cu:157-161 */
            radiance = vs(mat.emission);
            break;
        }

        /* device.cu:164-173 */
        float bx = h.u, by = h.v;
        float bw = 1.0f - bx - by;
        const float* P = s->gpos + (size_t)prim * 9;
        const float* N = s->nrm + (size_t)prim * 9;
        v3 v_p = V(dm_fma(by, P[6], dm_fma(bx, P[3], bw * P[0])), dm_fma(by, P[7], dm_fma(bx, P[4], bw * P[1])),
                   dm_fma(by, P[8], dm_fma(bx, P[5], bw * P[2])));
        v3 v_n = vnormalize(V(dm_fma(by, N[6], dm_fma(bx, N[3], bw * N[0])), dm_fma(by, N[7], dm_fma(bx, N[4], bw * N[1])),
                              dm_fma(by, N[8], dm_fma(bx, N[5], bw * N[2]))));
        if (s->tex_idx[prim] >= 0 && s->tc) { /* device.cu:75-94 */
            const float* C = s->tc + (size_t)prim * 6;
            float tu = dm_fma(by, C[4], dm_fma(bx, C[2], bw * C[0]));
            float tv = dm_fma(by, C[5], dm_fma(bx, C[3], bw * C[1]));
            mat.base_color = tex_nearest(&s->textures[s->tex_idx[prim]], tu, tv);
        }
        if (cnt) cnt->scatters++;

        /* device.cu:176-190 */
        v3 T, B;
        onb(v_n, &T, &B);
        v3 local_wo = to_local(T, B, v_n, wo);
        v3 local_wi = vs(0.0f);
        float pdf = 0.0f;
        if (lg) { lg[11] = bits_f((uint32_t)mi); lg[12] = bits_f(*rng); st3(lg + 13, local_wo); }
        v3 f = sample_disney(&mat, local_wo, rng, &local_wi, &pdf, &sampled_lobe);
        v3 wi = to_world(T, B, v_n, local_wi);
        if (lg) { st3(lg + 16, f); lg[19] = pdf; st3(lg + 20, local_wi); lg[23] = bits_f((uint32_t)sampled_lobe); lg[27] = bits_f(*rng); }

        if (pdf < 1e-5f) break; /* device.cu:193 */

        if (dm_isinf(f.x) || dm_isinf(f.y) || dm_isinf(f.z) || dm_isnan(f.x) || dm_isnan(f.y) || dm_isnan(f.z)) {
            /* device.cu:196-201: --depth; continue; -> the SAME ray is traced again with fresh RNG draws */
            if (cnt) cnt->nan_retries++;
            /* safety net shared with the HIP kernel: the reference would spin forever on a hit whose
               BSDF is NaN for every draw; after 64 consecutive retries the path ends with zero radiance */
            if (++retries > 64) break;
            --depth;
            continue;
        }
        retries = 0;

        /* device.cu:204-205 */
        float aci = dm_abs(cos_theta(local_wi));
        throughput = vmul(throughput, vdivs(vscale(f, aci), pdf));
        org = v_p;
        dir = wi;
        if (lg) st3(lg + 24, throughput);

        /* device.cu:209-214: inverted, uncompensated Russian roulette */
        float beta_max = dm_max(throughput.x, dm_max(throughput.y, throughput.z));
        if (sampled_lobe != LOBE_GLASS && depth > 3) {
            float q = dm_max(0.05f, 1.0f - beta_max);
            if (rng_next(rng) > q) break;
        }
    }
    return vmul(radiance, throughput); /* device.cu:217 */
}

/* one pixel of ray_gen, device.cu:220-254 */
static v3 render_pixel(const render_ctx* rc, const orc_camera* cam, int W, int H, int px, int py, int max_samples, orc_counters* cnt,
                       float* per_sample_rgb, uint32_t* per_sample_state)
{
    uint32_t rng = orc_rng_init((uint32_t)px, (uint32_t)py);
    v3 color = vs(0.0f);
    v3 origin = ld3(cam->origin), llc = ld3(cam->llc), hor = ld3(cam->horizontal), ver = ld3(cam->vertical);
    for (int s = 0; s < max_samples; ++s) {
        float rx = rng_next(&rng);
        float ry = rng_next(&rng);
        float su = ((float)px + rx) / (float)W;
        float sv = ((float)py + ry) / (float)H;
        /* llc + u*horizontal + v*vertical - origin, left to right (device.cu:239-240) */
        v3 d = vsub(vadd(vadd(llc, vscale(hor, su)), vscale(ver, sv)), origin);
        d = vnormalize(d);
        v3 r = trace_path(rc, origin, d, &rng, cnt);
        color = vadd(color, r);
        if (cnt) cnt->samples++;
        if (per_sample_rgb) st3(per_sample_rgb + (size_t)s * 3, r);
        if (per_sample_state) per_sample_state[s] = rng;
    }
    color = vscale(color, 1.0f / (float)max_samples); /* device.cu:247 */
    return color;
}

typedef struct job {
    render_ctx rc;
    const orc_camera* cam;
    int W, H, max_samples;
    const uint32_t* pixel_list;
    int64_t n_pixels;
    float* out_rgb;
    uint32_t* out_rgba8;
    int64_t next;
    pthread_mutex_t mu;
    orc_counters total;
    int want_counters;
} job;

static void* worker(void* arg)
{
    job* j = (job*)arg;
    orc_counters local;
    memset(&local, 0, sizeof(local));
    const int64_t chunk = 256;
    for (;;) {
        int64_t begin = __atomic_fetch_add(&j->next, chunk, __ATOMIC_RELAXED);
        if (begin >= j->n_pixels) break;
        int64_t end = begin + chunk < j->n_pixels ? begin + chunk : j->n_pixels;
        for (int64_t i = begin; i < end; ++i) {
            uint32_t id = j->pixel_list ? j->pixel_list[i] : (uint32_t)i;
            int px = (int)(id % (uint32_t)j->W), py = (int)(id / (uint32_t)j->W);
            v3 c = render_pixel(&j->rc, j->cam, j->W, j->H, px, py, j->max_samples, j->want_counters ? &local : NULL, NULL, NULL);
            size_t ofs = (size_t)px + (size_t)j->W * (size_t)(j->H - 1 - py); /* device.cu:251 */
            if (j->out_rgb) st3(j->out_rgb + ofs * 3, c);
            if (j->out_rgba8) j->out_rgba8[ofs] = make_rgba(c);
        }
    }
    if (j->want_counters) {
        pthread_mutex_lock(&j->mu);
        j->total.rays += local.rays; j->total.nodes += local.nodes; j->total.tris += local.tris;
        j->total.scatters += local.scatters; j->total.env_misses += local.env_misses; j->total.samples += local.samples;
        j->total.nan_retries += local.nan_retries;
        pthread_mutex_unlock(&j->mu);
    }
    return NULL;
}

int orc_render(const orc_scene* s, const orc_camera* cam, const orc_env* env, int W, int H, int max_samples, int max_depth, int use_bvh,
               int n_threads, const uint32_t* pixel_list, int64_t n_pixels, float* out_rgb, uint32_t* out_rgba8, orc_counters* counters)
{
    if (!s || !cam || !env || W <= 0 || H <= 0 || max_samples <= 0) return -1;
    job j;
    memset(&j, 0, sizeof(j));
    j.rc.s = s; j.rc.env = env; j.rc.max_depth = max_depth; j.rc.use_bvh = use_bvh;
    j.cam = cam; j.W = W; j.H = H; j.max_samples = max_samples;
    j.pixel_list = pixel_list;
    j.n_pixels = pixel_list ? n_pixels : (int64_t)W * H;
    j.out_rgb = out_rgb; j.out_rgba8 = out_rgba8;
    j.want_counters = counters != NULL;
    pthread_mutex_init(&j.mu, NULL);
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    for (int i = 1; i < n_threads; ++i) pthread_create(&th[i], NULL, worker, &j);
    worker(&j);
    for (int i = 1; i < n_threads; ++i) pthread_join(th[i], NULL);
    pthread_mutex_destroy(&j.mu);
    if (counters) *counters = j.total;
    return 0;
}

/* Test infrastructure for bisecting a parity failure: the per-bounce log (see ORC_LOG_FLOATS) of sample `sample` of pixel (px, py).
   Returns the number of rows written. */
int orc_trace_sample(const orc_scene* s, const orc_camera* cam, const orc_env* env, int W, int H, int px, int py, int sample, int max_depth,
                     int use_bvh, float* log, int max_rows)
{
    render_ctx rc;
    rc.s = s; rc.env = env; rc.max_depth = max_depth; rc.use_bvh = use_bvh;
    uint32_t rng = orc_rng_init((uint32_t)px, (uint32_t)py);
    v3 origin = ld3(cam->origin), llc = ld3(cam->llc), hor = ld3(cam->horizontal), ver = ld3(cam->vertical);
    int rows = 0;
    for (int k = 0; k <= sample; ++k) {
        float rx = rng_next(&rng);
        float ry = rng_next(&rng);
        float su = ((float)px + rx) / (float)W;
        float sv = ((float)py + ry) / (float)H;
        v3 d = vnormalize(vsub(vadd(vadd(llc, vscale(hor, su)), vscale(ver, sv)), origin));
        if (k == sample) { g_log = log; g_log_rows = 0; g_log_cap = max_rows; }
        trace_path(&rc, origin, d, &rng, NULL);
        if (k == sample) { rows = g_log_rows; g_log = NULL; g_log_cap = 0; }
    }
    return rows;
}

int orc_trace_pixel(const orc_scene* s, const orc_camera* cam, const orc_env* env, int W, int H, int px, int py, int max_samples,
                    int max_depth, int use_bvh, float* per_sample_rgb, uint32_t* per_sample_state)
{
    render_ctx rc;
    rc.s = s; rc.env = env; rc.max_depth = max_depth; rc.use_bvh = use_bvh;
    render_pixel(&rc, cam, W, H, px, py, max_samples, NULL, per_sample_rgb, per_sample_state);
    return 0;
}
