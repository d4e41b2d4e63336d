/* abi_conformance.c -- a COMPILED consumer of include/threecrate_hip.h (test infrastructure).
 *
 * The product's host side is meant to be bound from Rust; this image has no Rust toolchain, so this
 * plain-C11 program (also compiled as C++) is the closest stand-in: it includes the public header, nothing else of the
 * repository, and
 *   layout            prints sizeof / offsetof of every public struct and the value of every public constant, one
 *                     `name value` pair per line -- tests/test_abi_conformance.py compares them with the ctypes mirror
 *                     (threecrate_amd/_lib.py) and with the #[repr(C)] structs of bindings/rust;
 *   run IN OUT        reads a scan pair from IN, calls the HOST entry points a drop-in caller would
 *                     (tc_estimate_normals -> estimate_normals, normals.rs:238-247; tc_icp_point_to_plane_detailed ->
 *                     icp_point_to_plane_detailed, registration.rs:508-516; tc_icp_detailed -> icp_detailed, :258-265;
 *                     tc_icp -> icp, :232-237) and writes the raw results to OUT; the pytest compares them with the
 *                     golden fixtures (GPU box only).
 * Build: gcc -std=c11 -Wall -Wextra -Werror -Iinclude tests/abi/abi_conformance.c -Lthreecrate_amd -lthreecrate_hip
 */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "threecrate_hip.h"

#define SZ(T) printf("sizeof." #T " %zu\n", sizeof(T))
#define OFF(T, f) printf("offsetof." #T "." #f " %zu\n", offsetof(T, f))
#define VAL(c) printf("const." #c " %lld\n", (long long)(c))

static int layout(void) {
    SZ(tc_normal_config);
    OFF(tc_normal_config, k_neighbors); OFF(tc_normal_config, radius); OFF(tc_normal_config, has_radius);
    OFF(tc_normal_config, consistent_orientation); OFF(tc_normal_config, has_viewpoint); OFF(tc_normal_config, viewpoint);
    SZ(tc_icp_result);
    OFF(tc_icp_result, transformation); OFF(tc_icp_result, mse); OFF(tc_icp_result, iterations); OFF(tc_icp_result, converged);
    OFF(tc_icp_result, n_correspondences); OFF(tc_icp_result, corr_target);
    SZ(tc_batch_icp_job);
    OFF(tc_batch_icp_job, source); OFF(tc_batch_icp_job, n_source); OFF(tc_batch_icp_job, target); OFF(tc_batch_icp_job, n_target);
    OFF(tc_batch_icp_job, max_iterations); OFF(tc_batch_icp_job, convergence_threshold); OFF(tc_batch_icp_job, max_correspondence_distance);
    SZ(tc_batch_icp_result);
    OFF(tc_batch_icp_result, transformation); OFF(tc_batch_icp_result, final_error); OFF(tc_batch_icp_result, iterations);
    OFF(tc_batch_icp_result, status);
    SZ(tc_kernel_stat);
    OFF(tc_kernel_stat, name); OFF(tc_kernel_stat, launches); OFF(tc_kernel_stat, total_ms); OFF(tc_kernel_stat, min_ms);
    OFF(tc_kernel_stat, max_ms);
    SZ(tc_icp_scale_level);
    OFF(tc_icp_scale_level, voxel_size); OFF(tc_icp_scale_level, max_iterations); OFF(tc_icp_scale_level, max_correspondence_distance);
    SZ(tc_multiscale_icp_config);
    OFF(tc_multiscale_icp_config, levels); OFF(tc_multiscale_icp_config, n_levels); OFF(tc_multiscale_icp_config, final_refinement_iterations);
    OFF(tc_multiscale_icp_config, final_max_correspondence_distance); OFF(tc_multiscale_icp_config, convergence_threshold);
    SZ(tc_gicp_config);
    OFF(tc_gicp_config, max_iterations); OFF(tc_gicp_config, max_correspondence_distance); OFF(tc_gicp_config, convergence_threshold);
    OFF(tc_gicp_config, k_correspondences);
    SZ(tc_kiss_icp_config);
    OFF(tc_kiss_icp_config, voxel_size); OFF(tc_kiss_icp_config, max_range); OFF(tc_kiss_icp_config, min_range);
    OFF(tc_kiss_icp_config, max_iterations);
    SZ(tc_frame_stream_config);
    OFF(tc_frame_stream_config, max_points); OFF(tc_frame_stream_config, max_queue_depth); OFF(tc_frame_stream_config, voxel_size);
    OFF(tc_frame_stream_config, k_neighbors); OFF(tc_frame_stream_config, max_iterations);
    OFF(tc_frame_stream_config, max_correspondence_distance); OFF(tc_frame_stream_config, convergence_threshold);
    SZ(tc_frame_result);
    OFF(tc_frame_result, transformation); OFF(tc_frame_result, mse); OFF(tc_frame_result, iterations); OFF(tc_frame_result, converged);
    OFF(tc_frame_result, status); OFF(tc_frame_result, n_points_in); OFF(tc_frame_result, n_points);
    SZ(tc_frame_stream_metrics);
    OFF(tc_frame_stream_metrics, items_queued); OFF(tc_frame_stream_metrics, items_processed); OFF(tc_frame_stream_metrics, items_dropped);
    OFF(tc_frame_stream_metrics, max_depth_seen);
    VAL(TC_ABI_VERSION); VAL(TC_OK); VAL(TC_INVALID_DATA); VAL(TC_ALGORITHM); VAL(TC_GPU); VAL(TC_UNSUPPORTED);
    VAL(TC_ICP_SUMS_P2PLANE); VAL(TC_ICP_SUMS_P2P); VAL(TC_ICP_SUMS_STRIDE); VAL(TC_COMM_ID_BYTES);
    VAL(TC_COLL_SUM_F64); VAL(TC_COLL_SUM_U32); VAL(TC_COLL_ALLGATHER_U8); VAL(TC_SHARD_SPATIAL); VAL(TC_SHARD_LOCAL); VAL(TC_SHARD_INDEX);
    VAL(TC_COUNTER_INDEXED_POINTS); VAL(TC_COUNTER_INDEX_BUILDS);
    VAL(TC_COUNTER_ICP_ITERATIONS); VAL(TC_COUNTER_ICP_TRIPS); VAL(TC_COUNTER_ICP_TRIPS_WITHOUT_SEARCH); VAL(TC_COUNTER_ICP_SEARCHES);
    VAL(TC_COUNTER_ICP_STEPS_NEEDED); VAL(TC_COUNTER_ICP_STEPS_TAKEN);
    /* the library this program is linked against answers for itself (no device needed) */
    printf("call.tc_abi_version %d\n", tc_abi_version());
    return 0;
}

static void *must_alloc(size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (!p) { fprintf(stderr, "out of memory\n"); exit(3); }
    return p;
}

#define CHECK(call)                                                                                    \
    do {                                                                                               \
        tc_status st_ = (call);                                                                        \
        if (st_ != TC_OK) {                                                                            \
            fprintf(stderr, "%s -> %d: %s\n", #call, (int)st_, ctx ? tc_last_error_message(ctx) : ""); \
            return 2;                                                                                  \
        }                                                                                              \
    } while (0)

static void put_result(FILE *f, const tc_icp_result *r) {
    /* 7 floats, mse, then iterations / converged / n_correspondences as u64 */
    uint64_t tail[3];
    fwrite(r->transformation, sizeof(float), 7, f);
    fwrite(&r->mse, sizeof(float), 1, f);
    tail[0] = r->iterations; tail[1] = (uint64_t)(r->converged != 0); tail[2] = r->n_correspondences;
    fwrite(tail, sizeof(uint64_t), 3, f);
}

static int run(const char *in_path, const char *out_path) {
    tc_context *ctx = NULL;
    uint64_t hdr[4];       /* n_source, n_target, k_neighbors, icp iterations */
    FILE *f = fopen(in_path, "rb");
    if (!f || fread(hdr, sizeof(uint64_t), 4, f) != 4) { fprintf(stderr, "cannot read %s\n", in_path); return 3; }
    const size_t ns = (size_t)hdr[0], nt = (size_t)hdr[1], k = (size_t)hdr[2], iters = (size_t)hdr[3];
    float *tgt = (float *)must_alloc(nt * 3 * sizeof(float)), *src = (float *)must_alloc(ns * 3 * sizeof(float));
    if (fread(tgt, sizeof(float), nt * 3, f) != nt * 3 || fread(src, sizeof(float), ns * 3, f) != ns * 3) { fprintf(stderr, "short input\n"); return 3; }
    fclose(f);

    CHECK(tc_context_create(0, &ctx));
    /* estimate_normals(&cloud, k) */
    tc_normal_config cfg;
    tc_normal_config_default(&cfg);
    if (cfg.k_neighbors != 10 || cfg.consistent_orientation != 1 || cfg.has_radius || cfg.has_viewpoint) { fprintf(stderr, "defaults differ from normals.rs:28-36\n"); return 2; }
    cfg.k_neighbors = k;
    float *np6 = (float *)must_alloc(nt * 6 * sizeof(float));
    CHECK(tc_estimate_normals(ctx, tgt, nt, &cfg, np6));
    /* icp_point_to_plane_detailed(&source, &target, &normals, init, iters, None, 0.0): the normals straight out of the
       NormalPoint3f array (stride 6) */
    const float identity[7] = {0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f};
    uint32_t *corr_pl = (uint32_t *)must_alloc(ns * sizeof(uint32_t)), *corr_pp = (uint32_t *)must_alloc(ns * sizeof(uint32_t));
    tc_icp_result pl, pp;
    memset(&pl, 0, sizeof pl); memset(&pp, 0, sizeof pp);
    pl.corr_target = corr_pl; pp.corr_target = corr_pp;
    CHECK(tc_icp_point_to_plane_detailed(ctx, src, ns, tgt, nt, np6 + 3, nt, 6, identity, iters, -1.0f, 0.0f, &pl));
    /* icp_detailed(&source, &target, init, iters, None, 0.0) */
    CHECK(tc_icp_detailed(ctx, src, ns, tgt, nt, identity, iters, -1.0f, 0.0f, &pp));
    /* icp(&source, &target, init, iters) -> Isometry3 */
    float t_icp[7];
    CHECK(tc_icp(ctx, src, ns, tgt, nt, identity, iters, t_icp));
    /* the reference's validation, through the compiled ABI: k < 3 and a normals length mismatch are InvalidData, an empty
       cloud is Ok(empty) before the k check (normals.rs:261-269, registration.rs:522-526) */
    tc_normal_config bad = cfg;
    bad.k_neighbors = 2;
    const int st_k = (int)tc_estimate_normals(ctx, tgt, nt, &bad, np6), st_empty = (int)tc_estimate_normals(ctx, tgt, 0, &bad, np6);
    tc_icp_result dummy;
    memset(&dummy, 0, sizeof dummy);
    const int st_len = (int)tc_icp_point_to_plane_detailed(ctx, src, ns, tgt, nt, np6 + 3, nt - 1, 6, identity, iters, -1.0f, 0.0f, &dummy);
    CHECK(tc_estimate_normals(ctx, tgt, nt, &cfg, np6));      /* (np6 again: the k = 2 call must not have written it) */

    f = fopen(out_path, "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", out_path); return 3; }
    fwrite(np6, sizeof(float), nt * 6, f);
    put_result(f, &pl); fwrite(corr_pl, sizeof(uint32_t), ns, f);
    put_result(f, &pp); fwrite(corr_pp, sizeof(uint32_t), ns, f);
    fwrite(t_icp, sizeof(float), 7, f);
    const int32_t codes[3] = {st_k, st_empty, st_len};
    fwrite(codes, sizeof(int32_t), 3, f);
    fclose(f);
    tc_context_destroy(ctx);
    free(tgt); free(src); free(np6); free(corr_pl); free(corr_pp);
    return 0;
}

int main(int argc, char **argv) {
    if (argc == 2 && strcmp(argv[1], "layout") == 0) return layout();
    if (argc == 4 && strcmp(argv[1], "run") == 0) return run(argv[2], argv[3]);
    fprintf(stderr, "usage: %s layout | run IN OUT\n", argv[0]);
    return 64;
}
