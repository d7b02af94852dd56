/*
 * pt_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar double-precision CPU restatement of the reference path tracer's hot path
 * (JoarGunnarsson/Pathtracing), used only as the checker for the HIP library: by
 * tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg.  Nothing in
 * pathtracing_amd/ links, loads or calls it.
 *
 * It consumes the same flat PtkSceneDesc as the GPU library and can draw its random
 * numbers in two ways:
 *   ORACLE_RNG_COUNTER            the counter-based generator specified in include/ptk.h
 *                                 (what the GPU uses), so GPU and oracle trace the same
 *                                 paths and can be compared sample by sample;
 *   ORACLE_RNG_REFERENCE_STREAMS  the reference's own three sequential generators
 *                                 (std::minstd_rand + uniform_real_distribution,
 *                                 std::minstd_rand + normal_distribution, C rand();
 *                                 src/utils.cpp:11-30,120-121) with caller-given seeds,
 *                                 consumed in the reference's call order.  With the
 *                                 reference's generators re-seeded to the same values
 *                                 (oracle/ref_harness.cpp) a single-threaded reference
 *                                 render and this restatement agree bit for bit; that
 *                                 is how the restatement is pinned.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include "ptk.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_RNG_COUNTER = 0, ORACLE_RNG_REFERENCE_STREAMS = 1 };

typedef struct OracleRng {
    int32_t mode;
    uint32_t uniform_seed; /* reference streams: seed of the uniform minstd_rand */
    uint32_t normal_seed;  /*                    seed of the normal minstd_rand  */
    uint32_t rand_seed;    /*                    srand() seed                    */
} OracleRng;

typedef struct OracleScene OracleScene;

typedef struct OracleStats {
    uint64_t paths, extension_rays, shadow_rays, surface_vertices, scatter_vertices;
    uint64_t bvh_nodes_visited, triangles_tested;
    uint64_t vertices_by_material[PTK_MAT_COUNT];
} OracleStats;

const char* oracle_last_error(void);
int oracle_scene_create(const PtkSceneDesc* desc, OracleScene** out);
void oracle_scene_destroy(OracleScene* scene);
/* Number of triangles, and for primitive id k of a mesh object (index in the array the
 * BVH root sorted, src/bvh.cpp:143) the index of that triangle in OBJ face order. */
int oracle_mesh_face_index(const OracleScene* scene, int32_t object, int64_t primitive_id, int64_t* face_index);

/* find_closest_hit for one primary ray per pixel (AA as configured; with AA the jitter
 * draws come from `rng`).  Outputs as ptk_trace_primary. */
int oracle_trace_primary(OracleScene* scene, const PtkRenderJob* job, const OracleRng* rng, int32_t* object_index,
                         int32_t* primitive_index, double* distance, double* position, double* normal);
/* raytrace for every (pixel, sample): per-sample radiance / first-hit position / normal,
 * [pixel][sample][3]; pixels in index order, samples in order (the reference's loop order). */
int oracle_trace_samples(OracleScene* scene, const PtkRenderJob* job, const OracleRng* rng, double* color,
                         double* position, double* normal);
/* raytrace_section: per-pixel sample means folded into the buffers with increment_average. */
int oracle_render(OracleScene* scene, const PtkRenderJob* job, const OracleRng* rng, int32_t threads, double* image,
                  double* position, double* normal, uint64_t* counters);
int oracle_get_stats(OracleScene* scene, OracleStats* stats);

int oracle_denoise(const PtkDenoiseTask* tasks, int32_t n_tasks, int32_t width, int32_t height, double* image,
                   const double* position, const double* normal);

/* The counter generator itself, for direct tests of the ABI's RNG contract. */
uint64_t oracle_rng_key(uint64_t seed, uint32_t pixel, uint32_t sample);
double oracle_rng_uniform(uint64_t key, uint32_t counter);
uint64_t oracle_rng_fork(uint64_t key, uint32_t counter);

/* Pure helper functions, exported for known-answer tests against the reference's
 * (src/utils.cpp, src/materials.cpp). */
double oracle_fresnel_dielectric(double cos_incident, double n1, double n2);
double oracle_fresnel_conductor(double cos_theta, double n1, double k1, double n2, double k2);
double oracle_schlick_fresnel(double cos_incident, double n1, double n2);
double oracle_beckmann_D(const double half_vector[3], const double normal[3], double alpha);
double oracle_smith_G1(const double half_vector[3], const double normal[3], const double v[3], double alpha);
void oracle_rotate(const double p[3], double alpha, double beta, double gamma, double out[3]);
void oracle_refract(const double incident[3], const double normal[3], double eta, double out[3]);
int oracle_solve_quadratic(double b, double c, double* distance);

#ifdef __cplusplus
}
#endif
#endif
