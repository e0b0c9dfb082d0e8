// pt_device.h -- device-side math for the render megakernel (gfx950).
//
// Behavioural spec: the reference's header-only device code (path_tracer/src/{random,math,sample_methods}.hpp,
// path_tracer/src/device/disney/*.cuh); each function cites the lines it implements.
//
// Arithmetic contract (DESIGN.md "deterministic math"): IEEE binary32, no contraction (the file is built
// with -ffp-contract=off), fused multiply-add only where spelled __builtin_fmaf, correctly rounded
// division and sqrt (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), and NO ocml transcendental:
// sin/cos/tan/atan/atan2/asin/log/exp/pow are the bounded-domain polynomial routines below.  The same
// contract is restated independently by the CPU oracle, which is what makes bit-exact parity testable.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PTD __device__ __forceinline__

namespace ptd {

PTD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PTD float sqrt_(float x) { return __builtin_sqrtf(x); }
PTD float abs_(float x) { return __builtin_fabsf(x); }
// fminf/fmaxf semantics (NaN operand ignored) with a fixed answer for every input incl. signed zeros
PTD float min_(float a, float b) { return (b != b || a < b) ? a : b; }
PTD float max_(float a, float b) { return (b != b || a > b) ? a : b; }
PTD bool isinf_(float x) { return abs_(x) == __builtin_inff(); }
PTD bool isnan_(float x) { return x != x; }

constexpr float kPi = 3.14159265358979323f;        // types.hpp:9
constexpr float kTwoPi = 6.28318530717958648f;     // types.hpp:10
constexpr float kPiOverTwo = 1.57079632679489661f; // types.hpp:11
constexpr float kPiOverFour = 0.78539816339744830f;
constexpr float kInvPi = 0.31830988618379067f;
constexpr float kTMin = 1e-3f;     // types.hpp:16
constexpr float kTMax = 1e10f;     // types.hpp:17
constexpr float kAlphaMin = 0.001f; // types.hpp:18

// ---- deterministic libm -------------------------------------------------------------------------

PTD void sincos_(float x, float& s, float& c)
{
    float kf = (x * 0.63661975f + 12582912.0f) - 12582912.0f;
    int k = (int)kf;
    float r = fma_(kf, -1.5703125f, x);
    r = fma_(kf, -0.0004837512969970703f, r);
    r = fma_(kf, -7.549790126404332e-08f, r);
    float z = r * r;
    float sp = fma_(fma_(fma_(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    float cp = fma_(z * z, fma_(fma_(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                    fma_(-0.5f, z, 1.0f));
    float a = (k & 1) ? cp : sp;
    float b = (k & 1) ? sp : cp;
    s = (k & 2) ? -a : a;
    c = ((k + 1) & 2) ? -b : b;
}
PTD float tan_(float x) { float s, c; sincos_(x, s, c); return s / c; }

PTD float atan_(float x)
{
    float ax = abs_(x);
    float y0, t;
    if (ax > 2.414213562373095f) { y0 = kPiOverTwo; t = -(1.0f / ax); }
    else if (ax > 0.4142135623730950f) { y0 = kPiOverFour; t = (ax - 1.0f) / (ax + 1.0f); }
    else { y0 = 0.0f; t = ax; }
    float z = t * t;
    float q = fma_(fma_(fma_(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
    float y = y0 + fma_(q * z, t, t);
    return (x < 0.0f) ? -y : y;
}

PTD float atan2_(float y, float x)
{
    if (x == 0.0f) {
        if (y == 0.0f) return 0.0f;
        return (y > 0.0f) ? kPiOverTwo : -kPiOverTwo;
    }
    float a = atan_(y / x);
    if (x < 0.0f) a = (y < 0.0f) ? a - kPi : a + kPi;
    return a;
}

PTD float asin_(float x)
{
    float a = abs_(x);
    if (a > 1.0f) return __builtin_nanf("");
    float z, w;
    bool big = a > 0.5f;
    if (big) { z = 0.5f * (1.0f - a); w = sqrt_(z); }
    else { w = a; z = w * w; }
    float p = fma_(fma_(fma_(fma_(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z, 1.6666752422e-1f);
    float r = fma_(p * z, w, w);
    if (big) r = kPiOverTwo - (r + r);
    return (x < 0.0f) ? -r : r;
}

PTD float log_poly_(float m)
{
    float p = fma_(7.0376836292e-2f, m, -1.1514610310e-1f);
    p = fma_(p, m, 1.1676998740e-1f);
    p = fma_(p, m, -1.2420140846e-1f);
    p = fma_(p, m, 1.4249322787e-1f);
    p = fma_(p, m, -1.6668057665e-1f);
    p = fma_(p, m, 2.0000714765e-1f);
    p = fma_(p, m, -2.4999993993e-1f);
    p = fma_(p, m, 3.3333331174e-1f);
    return p;
}

PTD float log_(float x)
{
    if (!(x > 0.0f)) return (x == 0.0f) ? -__builtin_inff() : __builtin_nanf("");
    if (x == __builtin_inff()) return x;
    uint32_t bits = __float_as_uint(x);
    int e = (int)((bits >> 23) & 0xffu) - 126;
    float m = __uint_as_float((bits & 0x807fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; }
    else { m = m - 1.0f; }
    float z = m * m;
    float p = log_poly_(m);
    float y = (p * m) * z;
    float fe = (float)e;
    y = fma_(-2.12194440e-4f, fe, y);
    y = fma_(-0.5f, z, y);
    float r = m + y;
    r = fma_(0.693359375f, fe, r);
    return r;
}

PTD float exp_(float x)
{
    if (x != x) return x;
    if (x > 88.72283905206835f) return __builtin_inff();
    if (x < -103.278929903431851103f) return 0.0f;
    float nf = (x * 1.44269504088896341f + 12582912.0f) - 12582912.0f;
    int n = (int)nf;
    float r = fma_(nf, -0.693359375f, x);
    r = fma_(nf, 2.12194440e-4f, r);
    float z = r * r;
    float p = fma_(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fma_(p, r, 8.3334519073e-3f);
    p = fma_(p, r, 4.1665795894e-2f);
    p = fma_(p, r, 1.6666665459e-1f);
    p = fma_(p, r, 5.0000001201e-1f);
    float y = fma_(p, z, r) + 1.0f;
    int n1 = n / 2, n2 = n - n1;
    float s1 = __uint_as_float((uint32_t)(n1 + 127) << 23);
    float s2 = __uint_as_float((uint32_t)(n2 + 127) << 23);
    return (y * s1) * s2;
}

PTD void log_ext_(float x, float& hi, float& lo)
{
    uint32_t bits = __float_as_uint(x);
    int e = (int)((bits >> 23) & 0xffu) - 126;
    float m = __uint_as_float((bits & 0x807fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; }
    else { m = m - 1.0f; }
    float z = m * m;
    float p = log_poly_(m);
    float fe = (float)e;
    float c = (p * m) * z;
    c = fma_(-2.12194440e-4f, fe, c);
    float zl = fma_(m, m, -z);
    c = fma_(-0.5f, zl, c);
    float a = 0.693359375f * fe;
    float s = a + m;
    float bb = s - a;
    float s_lo = (a - (s - bb)) + (m - bb);
    float hz = -0.5f * z;
    float s2 = s + hz;
    float bb2 = s2 - s;
    float s2_lo = (s - (s2 - bb2)) + (hz - bb2);
    float t = (s_lo + s2_lo) + c;
    float h = s2 + t;
    hi = h;
    lo = (s2 - h) + t;
}

// x >= 0 only
PTD float pow_(float x, float y)
{
    if (y == 0.0f) return 1.0f;
    if (x == 1.0f) return 1.0f;
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : __builtin_inff();
    if (!(x > 0.0f) || x == __builtin_inff() || x < 1.17549435e-38f) return exp_(y * log_(x));
    float lh, ll;
    log_ext_(x, lh, ll);
    float ph = y * lh;
    float pl = fma_(y, lh, -ph) + y * ll;
    float eh = exp_(ph);
    return fma_(eh, pl, eh);
}

// ---- vec3 ---------------------------------------------------------------------------------------

struct v3 {
    float x, y, z;
};
PTD v3 V(float x, float y, float z) { return v3{x, y, z}; }
PTD v3 vs(float s) { return v3{s, s, s}; }
PTD v3 operator+(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
PTD v3 operator-(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
PTD v3 operator*(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
PTD v3 operator*(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
PTD v3 operator/(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
PTD v3 operator-(v3 a) { return V(-a.x, -a.y, -a.z); }
PTD float dot(v3 a, v3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
PTD v3 cross(v3 a, v3 b)
{
    return V(fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x)));
}
PTD v3 normalize(v3 a) { return a * (1.0f / sqrt_(dot(a, a))); } // owl::normalize = v * rsqrt(dot(v,v))

PTD float lerpf(float a, float b, float t) { return fma_(b - a, t, a); } // math.hpp:6
PTD v3 lerp3(v3 a, v3 b, float t) { return V(lerpf(a.x, b.x, t), lerpf(a.y, b.y, t), lerpf(a.z, b.z, t)); } // math.hpp:10
PTD float sqr(float v) { return v * v; } // math.hpp:16

// math.hpp:22-38
PTD float cos_theta(v3 w) { return w.z; }
PTD float sin_theta(v3 w) { return sqrt_(max_(0.0f, 1.0f - sqr(cos_theta(w)))); }
PTD float tan_theta(v3 w) { return sin_theta(w) / cos_theta(w); }
PTD float clampf(float x, float lo, float hi) { return min_(hi, max_(lo, x)); } // disney_helper.cuh:14-17
PTD float cos_phi(v3 w)
{
    float theta = sin_theta(w);
    return (theta == 0.0f) ? 1.0f : clampf(w.x / theta, -1.0f, 1.0f);
}
PTD float sin_phi(v3 w)
{
    float theta = sin_theta(w);
    return (theta == 0.0f) ? 1.0f : clampf(w.y / theta, -1.0f, 1.0f); // 1 at the pole, math.hpp:34-38
}

// math.hpp:50-56 / :42-48
PTD v3 to_sphere3(float sin_t, float cos_t, float phi)
{
    float s, c;
    sincos_(phi, s, c);
    return V(sin_t * c, sin_t * s, cos_t);
}
PTD v3 to_sphere2(float theta, float phi)
{
    float st, ct;
    sincos_(theta, st, ct);
    return to_sphere3(st, ct, phi);
}

PTD v3 reflect(v3 w, v3 n) { return (n * dot(w, n)) * 2.0f - w; } // math.hpp:58-61

// math.hpp:63-77
PTD bool refract(v3 w, v3 n, float eta, v3& wi)
{
    if (eta == 1.0f) { wi = -w; return true; }
    float cos_theta_i = dot(w, n);
    float sin2_theta_i = max_(0.0f, 1.0f - sqr(cos_theta_i));
    float sin2_theta_t = eta * eta * sin2_theta_i;
    if (sin2_theta_t > 1.0f) return false;
    float cos_theta_t = sqrt_(1.0f - sin2_theta_t);
    wi = (-w) * eta + n * (eta * cos_theta_i - cos_theta_t);
    return true;
}
PTD bool same_hemisphere(v3 a, v3 b) { return a.z * b.z > 0.0f; } // math.hpp:79-82

// math.hpp:86-95
PTD void onb(v3 n, v3& t, v3& b)
{
    if (n.x != n.y || n.x != n.z) t = V(n.z - n.y, n.x - n.z, n.y - n.x);
    else t = V(n.z - n.y, n.x + n.z, -n.y - n.x);
    t = normalize(t);
    b = cross(n, t);
}
PTD v3 to_local(v3 t, v3 b, v3 n, v3 w) { return normalize(V(dot(w, t), dot(w, b), dot(w, n))); } // math.hpp:98-101
PTD v3 to_world(v3 t, v3 b, v3 n, v3 w) // math.hpp:104-107
{
    v3 r = V(fma_(w.z, n.x, fma_(w.y, b.x, w.x * t.x)), fma_(w.z, n.y, fma_(w.y, b.y, w.x * t.y)),
             fma_(w.z, n.z, fma_(w.y, b.z, w.x * t.z)));
    return normalize(r);
}

// ---- RNG: random.hpp:34-85 ----------------------------------------------------------------------

PTD uint32_t rng_init(uint32_t seed_u, uint32_t seed_v) // random.hpp:46-56
{
    uint32_t s = 0;
#pragma unroll
    for (int n = 0; n < 4; n++) {
        s += 0x9e3779b9u;
        seed_u += ((seed_v << 4) + 0xa341316cu) ^ (seed_v + s) ^ ((seed_v >> 5) + 0xc8013ea4u);
        seed_v += ((seed_u << 4) + 0xad90777du) ^ (seed_u + s) ^ ((seed_u >> 5) + 0x7e95761eu);
    }
    return seed_u;
}
PTD float rng_next(uint32_t& state) // random.hpp:61-69; may return exactly 1.0f
{
    state = 16807u * state + 1013904223u;
    return (float)state * 0x1p-32f;
}

// ---- sampling: sample_methods.hpp ---------------------------------------------------------------

PTD void sample_concentric_disk(float rx, float ry, float& ox, float& oy) // sample_methods.hpp:19-41
{
    float dx = 2.0f * rx - 1.0f;
    float dy = 2.0f * ry - 1.0f;
    if (dx == 0.0f && dy == 0.0f) { ox = 0.0f; oy = 0.0f; return; }
    float phi, r;
    if (abs_(dx) > abs_(dy)) { r = dx; phi = kPiOverFour * (dy / dx); }
    else { r = dy; phi = kPiOverTwo - kPiOverFour * (dx / dy); }
    float s, c;
    sincos_(phi, s, c);
    ox = r * c;
    oy = r * s;
}
PTD v3 sample_cosine_hemisphere(float rx, float ry) // sample_methods.hpp:53-60
{
    float cx, cy;
    sample_concentric_disk(rx, ry, cx, cy);
    float ct = sqrt_(max_(0.0f, 1.0f - sqr(cx) - sqr(cy)));
    return V(cx, cy, ct);
}
PTD float pdf_cosine_hemisphere(v3 wi) { return abs_(cos_theta(wi)) * kInvPi; } // sample_methods.hpp:62-65

// ---- material + Disney BSDF ---------------------------------------------------------------------

struct Material { // material_data, device_global.hpp:19-36
    v3 base_color;
    float subsurface, metallic, specular, specular_tint, roughness, anisotropic, sheen, sheen_tint, clearcoat, clearcoat_gloss, ior,
        specular_transmission, specular_transmission_roughness, emission;
};
PTD Material material_default()
{
    return Material{{0.8f, 0.8f, 0.8f}, 0.0f, 0.0f, 0.5f, 1.0f, 0.5f, 0.0f, 0.0f, 1.0f, 0.0f, 0.03f, 1.45f, 0.0f, 0.0f, 0.0f};
}
template <typename P>
PTD Material material_load(P p)
{
    Material m;
    m.base_color = V(p[0], p[1], p[2]);
    m.subsurface = p[3]; m.metallic = p[4]; m.specular = p[5]; m.specular_tint = p[6]; m.roughness = p[7];
    m.anisotropic = p[8]; m.sheen = p[9]; m.sheen_tint = p[10]; m.clearcoat = p[11]; m.clearcoat_gloss = p[12];
    m.ior = p[13]; m.specular_transmission = p[14]; m.specular_transmission_roughness = p[15]; m.emission = p[16];
    return m;
}

PTD v3 rgb_to_lin(v3 c) { return V(pow_(c.x, 2.2f), pow_(c.y, 2.2f), pow_(c.z, 2.2f)); } // disney_helper.cuh:4-7
PTD float luminance(v3 c) { return dot(V(0.2126f, 0.7152f, 0.0722f), c); }                // disney_helper.cuh:9-12
PTD float schlick_weight(float ct)                                                        // disney_helper.cuh:19-24
{
    float m = clampf(1.0f - ct, 0.0f, 1.0f);
    float m2 = m * m;
    return m2 * m2 * m;
}
PTD float relative_eta(v3 wo, float ior, float& eta_i, float& eta_t) // disney_helper.cuh:31-37
{
    eta_i = cos_theta(wo) > 0.0f ? 1.0f : ior;
    eta_t = cos_theta(wo) > 0.0f ? ior : 1.0f;
    return eta_i / eta_t;
}
PTD float roughness_to_alpha1(float roughness) { return max_(kAlphaMin, clampf(sqr(roughness), 0.0f, 1.0f)); } // :39-42
PTD void roughness_to_alpha2(float roughness, float anisotropy, float& ax, float& ay)                          // :44-48
{
    float aspect = sqrt_(1.0f - 0.9f * anisotropy);
    ax = max_(kAlphaMin, sqr(roughness) / aspect);
    ay = max_(kAlphaMin, sqr(roughness) * aspect);
}
PTD float fresnel_equation(v3 i, v3 m, float eta_i, float eta_t) // disney_helper.cuh:52-60
{
    float c = abs_(dot(i, m));
    float denominator = sqr(eta_t / eta_i) - 1.0f + sqr(c);
    if (denominator < 0.0f) return 1.0f;
    float g = sqrt_(denominator);
    return 0.5f * sqr((g - c) / (g + c)) * (1.0f + sqr(c * (g + c) - 1.0f) / sqr(c * (g - c) + 1.0f));
}

PTD float lambda(v3 w, float ax, float ay) // disney_specular.cuh:17-27
{
    float abs_tan_theta = tan_theta(w);
    if (isinf_(abs_tan_theta)) return 0.0f;
    float alpha0 = sqrt_(sqr(cos_phi(w) * ax) + sqr(sin_phi(w) * ay));
    float a = 1.0f / (alpha0 * abs_tan_theta);
    return (-1.0f + sqrt_(1.0f + 1.0f / sqr(a))) / 2.0f;
}
PTD float g1_smith(v3 w, float ax, float ay) { return 1.0f / (1.0f + lambda(w, ax, ay)); } // :31-34
PTD float g2_smith_separable(v3 wo, v3 wi, float ax, float ay) { return g1_smith(wo, ax, ay) * g1_smith(wi, ax, ay); } // :38-41
PTD float g2_smith_correlated(v3 wo, v3 wi, float ax, float ay) { return 1.0f / (1.0f + lambda(wo, ax, ay) + lambda(wi, ax, ay)); } // :46-49
PTD float d_gtr_2(v3 wm, float ax, float ay) // disney_specular.cuh:54-60
{
    float tan2_theta = sqr(tan_theta(wm));
    if (isinf_(tan2_theta)) return 0.0f;
    float cos4_theta = sqr(sqr(cos_theta(wm)));
    float e = 1.0f + tan2_theta * (sqr(cos_phi(wm)) / sqr(ax) + sqr(sin_phi(wm)) / sqr(ay));
    return 1.0f / (kPi * ax * ay * cos4_theta * sqr(e));
}
PTD v3 sample_gtr2_ndf(float ax, float ay, float u0, float u1) // disney_specular.cuh:64-81 ("+ inv_pi" kept, :69)
{
    float phi = atan_(ay / ax * tan_(kTwoPi * u1 + kInvPi));
    if (u1 > 0.5f) phi += kPi;
    float sin_p, cos_p;
    sincos_(phi, sin_p, cos_p);
    float alphax2 = sqr(ax), alphay2 = sqr(ay);
    float alpha2 = 1.0f / (sqr(cos_p) / alphax2 + sqr(sin_p) / alphay2);
    float tan_theta2 = alpha2 * u0 / (1.0f - u0);
    float cos_t = 1.0f / sqrt_(1.0f + tan_theta2);
    float sin_t = sqrt_(max_(0.0f, 1.0f - sqr(cos_t)));
    return normalize(V(sin_t * cos_p, sin_t * sin_p, cos_t));
}

PTD v3 eval_disney_specular_brdf(const Material& m, v3 wo, v3 wh, v3 wi, float& pdf) // disney_specular.cuh:125-149
{
    float lum = luminance(m.base_color);
    v3 c_tint = lum > 0.0f ? m.base_color / lum : vs(1.0f);
    v3 c_spec = lerp3(lerp3(vs(1.0f), c_tint, m.specular_tint) * (0.08f * m.specular), m.base_color, m.metallic);
    float ax, ay;
    roughness_to_alpha2(m.roughness, m.anisotropic, ax, ay);
    float d = d_gtr_2(wh, ax, ay);
    float g = g2_smith_correlated(wo, wi, ax, ay);
    v3 f = lerp3(c_spec, vs(1.0f), schlick_weight(dot(wi, wh)));
    pdf = d * g1_smith(wo, ax, ay) * max_(0.0f, dot(wo, wh)) / (4.0f * cos_theta(wo)); // VNDF pdf for an NDF sample (:144)
    return (f * (d * g)) / (4.0f * abs_(cos_theta(wo)));                              // no 1/cos_i (:147-148)
}
PTD v3 sample_disney_specular_brdf(const Material& m, v3 wo, uint32_t& rng, v3& wi, float& pdf) // :151-170
{
    float ax, ay;
    roughness_to_alpha2(m.roughness, m.anisotropic, ax, ay);
    float u0 = rng_next(rng), u1 = rng_next(rng);
    v3 wh = sample_gtr2_ndf(ax, ay, u0, u1);
    if (dot(wo, wh) < 0.0f) wh = -wh;
    wi = reflect(wo, wh);
    if (cos_theta(wi) <= 0.0f) { pdf = 0.0f; return vs(0.0f); }
    return eval_disney_specular_brdf(m, wo, wh, wi, pdf);
}

PTD v3 sample_gtr2_bsdf(float a, float u0, float u1) // disney_specular.cuh:175-180
{
    float theta = atan_((a * sqrt_(u0)) / sqrt_(1.0f - u0));
    float phi = kTwoPi * u1;
    return to_sphere2(theta, phi);
}
PTD v3 eval_disney_specular_bsdf(const Material& m, v3 wo, v3 wh, v3 wi, float& pdf) // disney_specular.cuh:193-214
{
    float eta_i, eta_t;
    float eta = relative_eta(wo, m.ior, eta_i, eta_t);
    float R = fresnel_equation(wo, wh, eta_i, eta_t);
    float T = 1.0f - R;
    float pr = R, pt = T;
    if (same_hemisphere(wo, wi)) {
        pdf = pr / (pr + pt);
        return (m.base_color * R) / abs_(cos_theta(wi));
    }
    pdf = pt / (pr + pt);
    v3 sq = V(sqrt_(m.base_color.x), sqrt_(m.base_color.y), sqrt_(m.base_color.z));
    return ((sq * T) / abs_(cos_theta(wi))) / sqr(eta);
}
PTD v3 sample_disney_specular_bsdf(const Material& m, v3 wo, uint32_t& rng, v3& wi, float& pdf) // :216-244
{
    float u0 = rng_next(rng), u1 = rng_next(rng);
    v3 wh = sample_gtr2_bsdf(roughness_to_alpha1(m.specular_transmission_roughness), u0, u1);
    if (cos_theta(wo) < 0.0f && !same_hemisphere(wo, wh)) wh = -wh;
    float eta_i, eta_t;
    float eta = relative_eta(wo, m.ior, eta_i, eta_t);
    float R = fresnel_equation(wo, wh, eta_i, eta_t);
    float T = 1.0f - R;
    float pr = R, pt = T;
    bool resample = !refract(wo, wh, eta, wi);
    if (!resample) resample = rng_next(rng) < pr / (pr + pt); // short-circuit ||: third draw only if refraction succeeded (:235)
    if (resample) {
        float ax, ay;
        roughness_to_alpha2(m.roughness, m.anisotropic, ax, ay);
        float v0 = rng_next(rng), v1 = rng_next(rng);
        wh = sample_gtr2_ndf(ax, ay, v0, v1);
        wi = normalize(reflect(wo, wh));
    }
    return eval_disney_specular_bsdf(m, wo, wh, wi, pdf);
}

PTD float d_gtr1(v3 wh, float alpha) // disney_clearcoat.cuh:13-20
{
    if (alpha >= 1.0f) return kInvPi;
    float a2 = sqr(alpha);
    return (a2 - 1.0f) / (kPi * log_(a2) * (1.0f + (a2 - 1.0f) * sqr(cos_theta(wh))));
}
PTD v3 sample_gtr1_ndf(v3 wo, float a, float u0, float u1) // disney_clearcoat.cuh:23-33
{
    float alpha2 = sqr(a);
    float cos_t = sqrt_(max_(0.0f, (1.0f - pow_(alpha2, 1.0f - u0)) / (1.0f - alpha2)));
    float sin_t = sqrt_(max_(0.0f, 1.0f - sqr(cos_t)));
    float phi = kTwoPi * u1;
    v3 wh = to_sphere3(sin_t, cos_t, phi);
    if (!same_hemisphere(wo, wh)) wh = -wh;
    return wh;
}
PTD v3 eval_disney_clearcoat(const Material& m, v3 wo, v3 wh, v3 wi, float& pdf) // disney_clearcoat.cuh:45-59
{
    if (m.clearcoat <= 0.0f) { pdf = 0.0f; return vs(0.0f); }
    float d = d_gtr1(wh, lerpf(0.1f, 0.001f, m.clearcoat_gloss));
    float f = lerpf(1.0f, schlick_weight(cos_theta(wi)), 0.04f); // argument order as in the reference (:54)
    float g = g2_smith_separable(wo, wi, 0.25f, 0.25f);
    pdf = d / (4.0f * dot(wh, wi));
    return vs(d * g * f / (4.0f * abs_(cos_theta(wo)) * abs_(cos_theta(wi))));
}
PTD v3 sample_disney_clearcoat(const Material& m, v3 wo, uint32_t& rng, v3& wi, float& pdf) // disney_clearcoat.cuh:61-78
{
    float a = lerpf(0.1f, 0.001f, m.clearcoat_gloss);
    float u0 = rng_next(rng), u1 = rng_next(rng);
    v3 wh = sample_gtr1_ndf(wo, a, u0, u1);
    if (dot(wh, wo) < 0.0f) wh = -wh;
    wh = normalize(wh);
    wi = reflect(wo, wh);
    if (!same_hemisphere(wo, wi)) { pdf = 0.0f; return vs(0.0f); }
    return eval_disney_clearcoat(m, wo, wh, wi, pdf);
}

PTD v3 eval_disney_diffuse(const Material& m, v3 wo, v3 wi, float& pdf) // disney_diffuse.cuh:26-55
{
    float cos_theta_o = cos_theta(wo);
    float cos_theta_i = cos_theta(wi);
    float fresnel_o = schlick_weight(cos_theta_o);
    float fresnel_i = schlick_weight(cos_theta_i);
    v3 lambert = m.base_color * kInvPi;
    float fd = (1.0f - 0.5f * fresnel_o) * (1.0f - 0.5f * fresnel_i);
    float rr = m.roughness * (dot(wo, wi) + 1.0f);
    float fr = rr * (fresnel_i + fresnel_o + fresnel_o * fresnel_i * (rr - 1.0f));
    pdf = pdf_cosine_hemisphere(wi);
    return lambert * (fd + fr);
}
PTD v3 sample_disney_diffuse(const Material& m, v3 wo, uint32_t& rng, v3& wi, float& pdf) // disney_diffuse.cuh:57-62
{
    float u0 = rng_next(rng), u1 = rng_next(rng);
    wi = sample_cosine_hemisphere(u0, u1);
    return eval_disney_diffuse(m, wo, wi, pdf);
}

PTD v3 eval_disney_sheen(const Material& m, v3 wo, v3 wi) // disney_sheen.cuh:15-37
{
    if (m.sheen <= 0.0f) return vs(0.0f);
    v3 wh = wi + wo;
    if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return vs(0.0f);
    wh = normalize(wh);
    float lum = luminance(rgb_to_lin(m.base_color));
    float cos_theta_d = dot(wi, wh);
    v3 tint = (lum > 0.0f) ? m.base_color / lum : vs(1.0f);
    return (lerp3(vs(1.0f), tint, m.sheen_tint) * m.sheen) * schlick_weight(cos_theta_d);
}

constexpr int kLobeNone = -1, kLobeDiffuse = 0, kLobeClearcoat = 1, kLobeMetallic = 2, kLobeGlass = 3; // disney.cuh:9-13

PTD v3 sample_disney(const Material& m, v3 wo, uint32_t& rng, v3& wi, float& pdf, int& sampled_lobe) // disney.cuh:15-66
{
    float diffuse_weight = (1.0f - m.specular_transmission) * (1.0f - m.metallic);
    float metallic_weight = m.metallic;
    float clearcoat_weight = 0.25f * m.clearcoat;
    float glass_weight = (1.0f - m.metallic) * m.specular_transmission;
    float factor = 1.0f / (metallic_weight + glass_weight + diffuse_weight + clearcoat_weight);
    float p_metallic = metallic_weight * factor;
    float p_glass = glass_weight * factor;
    float p_diffuse = diffuse_weight * factor;
    float p_clearcoat = clearcoat_weight * factor;

    bool force_btdf = cos_theta(wo) < 0.0f && sampled_lobe == kLobeGlass;
    float p = rng_next(rng);
    v3 f = vs(0.0f);
    // thresholds in the reference's order metallic -> clearcoat -> diffuse -> glass, '<=' (disney.cuh:44-63)
    if (!force_btdf && p <= p_metallic) {
        f = sample_disney_specular_brdf(m, wo, rng, wi, pdf);
        sampled_lobe = kLobeMetallic;
    } else if (!force_btdf && p > p_metallic && p <= (p_metallic + p_clearcoat)) {
        f = sample_disney_clearcoat(m, wo, rng, wi, pdf);
        sampled_lobe = kLobeClearcoat;
    } else if (!force_btdf && p > p_metallic + p_clearcoat && p <= (p_metallic + p_clearcoat + p_diffuse)) {
        f = sample_disney_diffuse(m, wo, rng, wi, pdf);
        sampled_lobe = kLobeDiffuse;
    } else if (force_btdf || p_glass >= 0.0f) {
        f = sample_disney_specular_bsdf(m, wo, rng, wi, pdf);
        sampled_lobe = kLobeGlass;
    }
    return f + eval_disney_sheen(m, wo, wi); // pdf/f are the chosen lobe's only (disney.cuh:65)
}

// ---- framebuffer / textures ---------------------------------------------------------------------

PTD uint32_t make_8bit(float f) // owl::make_rgba (OWL source absent): min(255, max(0, int(f*256.f)))
{
    float s = f * 256.0f;
    int v = (s != s) ? 0 : (s >= 2147483520.0f ? 2147483647 : (s <= -2147483520.0f ? -2147483647 : (int)s));
    v = v < 0 ? 0 : v;
    v = v > 255 ? 255 : v;
    return (uint32_t)v;
}
PTD uint32_t make_rgba(v3 c) { return make_8bit(c.x) | (make_8bit(c.y) << 8) | (make_8bit(c.z) << 16) | (0xffu << 24); }

PTD int tex_coord(float u, int n)
{
    float fl = __builtin_floorf(u * (float)n);
    return (fl != fl) ? 0 : (fl >= (float)n ? n - 1 : (fl < 0.0f ? 0 : (int)fl));
}
// tex2D<float4> on an RGBA8 / nearest / clamp / normalised-coordinates texture (owl.hpp:248-257)
template <class TexelPtr>
PTD v3 tex_nearest(TexelPtr texels, int w, int h, float u, float v)
{
    int ix = tex_coord(u, w), iy = tex_coord(v, h);
    uint32_t p = texels[(size_t)iy * (size_t)w + (size_t)ix];
    return V((float)(p & 0xffu) / 255.0f, (float)((p >> 8) & 0xffu) / 255.0f, (float)((p >> 16) & 0xffu) / 255.0f);
}
PTD void uv_on_sphere(v3 n, float& u, float& v) // device.cu:23-28
{
    u = 0.5f + atan2_(n.x, n.z) / (2.0f * kPi);
    v = 0.5f + asin_(n.y) / kPi;
}

} // namespace ptd
