/*
 * pt_oracle.h -- CPU ORACLE for the path-tracing render loop.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm (jctemp/owl-path-tracer,
 * path_tracer/src/device/device.cu and the headers it includes).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * library (owl-path-tracer_amd/csrc) never includes, links or calls anything in oracle/.
 *
 * PARITY PINNED ONLY IN PART, UNPINNED ELSEWHERE: the reference ships no golden vectors, known-answer
 * tests or fixtures for this path (unit_tests/path_tracer_test.cu:10-31 is a placeholder) and it
 * cannot be built here (needs nvcc, OptiX 7.4, OWL, fmt, nlohmann, tinyobj -- all absent), so there
 * is no oracle/_ref.  What pins the oracle to the reference is what the reference itself rendered:
 * the furnace images under thesis/assets/furnace-test (tests/golden/furnace_reference.json, ring means
 * of the 8-bit values): camera, miss shader, diffuse lobe, the mirror limits of the specular and glass
 * lobes, the rough GGX lobe at normal incidence, the quantiser - every ring within +-1 code - and the
 * survey's RNG integer states (exact).  UNPINNED (no reference output exercises them): clearcoat, rough
 * glass, sheen, oblique rough specular, Russian roulette, textures, the traversal.  DESIGN.md 2.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAT_FLOATS 17 /* material_data, device_global.hpp:19-36 */

/* material_data field order (device_global.hpp:21-35) */
enum {
    ORC_M_BASE_R = 0, ORC_M_BASE_G, ORC_M_BASE_B, ORC_M_SUBSURFACE, ORC_M_METALLIC, ORC_M_SPECULAR,
    ORC_M_SPECULAR_TINT, ORC_M_ROUGHNESS, ORC_M_ANISOTROPIC, ORC_M_SHEEN, ORC_M_SHEEN_TINT,
    ORC_M_CLEARCOAT, ORC_M_CLEARCOAT_GLOSS, ORC_M_IOR, ORC_M_SPEC_TRANS, ORC_M_SPEC_TRANS_ROUGHNESS,
    ORC_M_EMISSION
};

typedef struct orc_texture {
    int32_t width, height;
    const uint32_t* rgba8; /* row 0 = v=0 (already flipped as application.cpp:229-234) */
} orc_texture;

/* Flattened scene: one record per triangle in global order (entity order, then face order). */
typedef struct orc_scene_desc {
    int32_t n_tris;
    const float* positions;    /* n_tris*9 : p0 p1 p2 */
    const float* normals;      /* n_tris*9 : n0 n1 n2 */
    const float* texcoords;    /* n_tris*6 : tc0 tc1 tc2 (may be NULL if no textures) */
    const int32_t* material_index; /* n_tris ; <0 => default material_data{} (device.cu:150-154) */
    const int32_t* texture_index;  /* n_tris ; <0 => has_texture=false */
    int32_t n_materials;
    const float* materials;    /* n_materials*17 */
    int32_t n_textures;
    const orc_texture* textures;
} orc_scene_desc;

typedef struct orc_env {
    int32_t use_map;   /* launch_params.environment_use */
    int32_t use_auto;  /* launch_params.environment_auto */
    float color[3];
    float intensity;
    orc_texture map;   /* width==0 => no map (texture object 0) */
} orc_env;

typedef struct orc_camera { /* camera_data, camera.hpp:14-20 */
    float origin[3], llc[3], horizontal[3], vertical[3];
} orc_camera;

typedef struct orc_counters {
    uint64_t rays, nodes, tris, scatters, env_misses, samples, nan_retries;
} orc_counters;

typedef struct orc_scene orc_scene;

orc_scene* orc_scene_create(const orc_scene_desc* d, int leaf_size);
void orc_scene_destroy(orc_scene* s);
void orc_scene_set_materials(orc_scene* s, const float* materials, int n_materials);
int orc_scene_bvh_nodes(const orc_scene* s);
int orc_scene_bvh_depth(const orc_scene* s);

/* camera.cpp:3-21 */
void orc_to_camera_data(const float look_from[3], const float look_at[3], const float look_up[3],
                        float vertical_fov, int w, int h, orc_camera* out);

/* Closest hit: returns 1 on hit. use_bvh=0 => brute force over all triangles. */
int orc_intersect(const orc_scene* s, const float org[3], const float dir[3], float tmin, float tmax,
                  int use_bvh, float* t, float* u, float* v, int32_t* prim);

/*
 * Render (ray_gen, device.cu:220-254).  out_rgb: W*H*3 floats, already row-flipped like the
 * reference framebuffer (index x + W*(H-1-y)); out_rgba8 optional (owl::make_rgba).
 * pixel_list (optional): launch-index pixels id = x + W*y to render; others are left untouched.
 * spp_begin/spp_count allow resuming: rng_state/accum (W*H u32 / W*H*3 float, launch-index order)
 * carry the per-pixel stream between chunks when non-NULL.
 */
int orc_render(const orc_scene* s, const orc_camera* cam, const orc_env* env, int W, int H,
               int max_samples, int max_depth, int use_bvh, int n_threads,
               const uint32_t* pixel_list, int64_t n_pixels,
               float* out_rgb, uint32_t* out_rgba8, orc_counters* counters);

/* Per-pixel trace for debugging parity: per-sample radiance (max_samples*3) and rng state after each sample. */
int orc_trace_pixel(const orc_scene* s, const orc_camera* cam, const orc_env* env, int W, int H, int px, int py,
                    int max_samples, int max_depth, int use_bvh, float* per_sample_rgb, uint32_t* per_sample_state);

/* Per-bounce log of ONE sample of one pixel (32 floats per bounce: origin, direction, hit flag, t u v, triangle, material, rng state before
   sample_disney, local wo, f, pdf, local wi, lobe, throughput, rng state after, radiance, depth - layout at ORC_LOG_FLOATS in pt_oracle.c).
   tools/fuzz_bisect.py replays the rows through the product's closest-hit and sample_disney debug ops to find where a differing sample
   parts ways.  Returns the number of rows written (<= max_rows). */
int orc_trace_sample(const orc_scene* s, const orc_camera* cam, const orc_env* env, int W, int H, int px, int py, int sample, int max_depth,
                     int use_bvh, float* log, int max_rows);

/* ---- unit hooks (vector tests) ---- */
uint32_t orc_rng_init(uint32_t seed_u, uint32_t seed_v);        /* random.hpp:46-56 */
float orc_rng_next(uint32_t* state);                            /* random.hpp:61-69 */
uint32_t orc_make_rgba(const float c[3]);                       /* owl::make_rgba (UNVERIFIED) */

/* sample_disney (disney.cuh:31-66): in/out lobe, rng state; returns f[3], wi[3], pdf */
void orc_sample_disney(const float mat[ORC_MAT_FLOATS], const float wo[3], uint32_t* rng_state, int32_t* sampled_lobe,
                       float f[3], float wi[3], float* pdf);
void orc_onb(const float n[3], float t[3], float b[3]);         /* math.hpp:86-95 */
void orc_to_local(const float t[3], const float b[3], const float n[3], const float w[3], float out[3]);
void orc_to_world(const float t[3], const float b[3], const float n[3], const float w[3], float out[3]);
void orc_sample_cosine_hemisphere(float u0, float u1, float out[3]);
int  orc_refract(const float w[3], const float n[3], float eta, float wi[3]);
float orc_fresnel_equation(const float i[3], const float m[3], float eta_i, float eta_t);
float orc_d_gtr1(const float wh[3], float alpha);
float orc_d_gtr2(const float wm[3], float ax, float ay);
float orc_lambda(const float w[3], float ax, float ay);
void orc_eval_lobe(int lobe, const float mat[ORC_MAT_FLOATS], const float wo[3], const float wh[3], const float wi[3],
                   float f[3], float* pdf);
void orc_eval_sheen(const float mat[ORC_MAT_FLOATS], const float wo[3], const float wi[3], float f[3]);
void orc_uv_on_sphere(const float n[3], float uv[2]);
void orc_tex_nearest(const orc_texture* t, float u, float v, float rgb[3]);

/* deterministic libm (bit-identical on host and gfx950; see DESIGN.md "deterministic math") */
float orc_dm_sin(float x);
float orc_dm_cos(float x);
float orc_dm_tan(float x);
float orc_dm_atan(float x);
float orc_dm_atan2(float y, float x);
float orc_dm_asin(float x);
float orc_dm_log(float x);
float orc_dm_exp(float x);
float orc_dm_pow(float x, float y);
void orc_dm_batch(int fn, const float* x, const float* y, float* out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
