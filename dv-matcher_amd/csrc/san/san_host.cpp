// Host-side sanitizer harness (test infrastructure): drives the C ABI's host code — argument validation, workspace
// sizing, arena carving, the context registry — under AddressSanitizer + UBSan.  No kernel is launched: every call
// either is a pure host function or is made with arguments that must be rejected before the launch (NULL pointers,
// bad sizes, a workspace one byte too small).  Run by tests/test_sanitizers.py on a machine without a GPU.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../../include/dvm.h"

static int fails = 0;
#define EXPECT(cond)                                                   \
    do {                                                               \
        if (!(cond)) {                                                 \
            printf("san_host: FAILED %s (line %d)\n", #cond, __LINE__); \
            ++fails;                                                   \
        }                                                              \
    } while (0)

int main() {
    EXPECT(dvm_abi_version() == DVM_ABI_VERSION);
    (void)dvm_device_count();
    // workspace queries: monotone in every size, never zero for real shapes, no overflow at the largest shipped shapes
    const int shapes[][3] = {{1, 20, 20}, {2, 256, 256}, {8, 2048, 2048}, {2, 4995, 2200}, {512, 2048, 2048}, {1, 8192, 8192}};
    size_t prev = 0;
    for (auto &sh : shapes) {
        const int B = sh[0], N = sh[1], M = sh[2];
        size_t a = dvm_softcorr_workspace_bytes(B, N, M, 128), b = dvm_pair_workspace_bytes(B, N, M),
               c = dvm_pair_direction_workspace_bytes(B, N, M), d = dvm_knn_neg_workspace_bytes(B, N, M, 128, 40),
               e = dvm_softcorr_bwd_workspace_bytes(B, N, M, 128), f = dvm_argmin_workspace_bytes(B, N, M, 128, 1),
               g = dvm_chamfer_workspace_bytes(B, N, M), h = dvm_dg_build_workspace_bytes(B, N), i = dvm_deformer_workspace_bytes(B, N, M, 10);
        EXPECT(a > 0 && b > 0 && c > 0 && d >= (size_t)B * N * M * 4 && e > 0 && f > 0 && g > 0 && h > 0 && i > 0);
        EXPECT(b >= c / 2);
        (void)prev;
        prev = b;
    }
    EXPECT(dvm_bn_workspace_bytes(8, 128, 2048) > 0 && dvm_sa_attention_workspace_bytes(8, 2048) > 0);
    EXPECT(dvm_pos_encoding_workspace_bytes() > 0 && dvm_proj2img_workspace_bytes(3) > 0);
    // argument validation: all of these must return DVM_EINVAL / DVM_ENOSPACE and set a message, touching nothing
    float dummy[64] = {0};
    int32_t idummy[64] = {0};
    EXPECT(dvm_softcorr_fwd_f32(nullptr, nullptr, 1, 8, 8, 128, -1.f, 10, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr) == DVM_EINVAL);
    EXPECT(strlen(dvm_last_error()) > 0);
    EXPECT(dvm_softcorr_fwd_f32(dummy, dummy, 1, 8, 8, 128, +1.f, 10, dummy, idummy, dummy, dummy, 0, dummy, 64, nullptr) == DVM_EINVAL);   // alpha sign
    EXPECT(dvm_softcorr_fwd_f32(dummy, dummy, 1, 8, 8, 128, -1.f, 99, dummy, idummy, dummy, dummy, 0, dummy, 64, nullptr) == DVM_EINVAL);   // topk range
    EXPECT(dvm_linear_f32(nullptr, dummy, 1, 4, 4, 4, 0, nullptr, nullptr, nullptr, nullptr, 1.f, dummy, nullptr) == DVM_EINVAL);
    EXPECT(dvm_linear_f32(dummy, dummy, 1, 4, 4, 4, 0, nullptr, nullptr, dummy, nullptr, 1.f, dummy, nullptr) == DVM_EINVAL);               // alpha without beta
    EXPECT(dvm_linear_f32(dummy, dummy, 1, 4, 100000, 4, 0, nullptr, nullptr, nullptr, nullptr, 1.f, dummy, nullptr) == DVM_EINVAL);        // K too large
    EXPECT(dvm_linear_prefix_f32(dummy, 6, dummy, dummy, 1, 4, 16, 4, nullptr, nullptr, nullptr, nullptr, 1.f, dummy, nullptr) == DVM_EINVAL);   // Cg % 4
    EXPECT(dvm_knn_neg_f32(dummy, dummy, 1, 4, 4, 4, 9, idummy, dummy, 64, nullptr) == DVM_EINVAL);                                        // k > M
    EXPECT(dvm_knn_neg_f32(dummy, dummy, 1, 4, 4, 4, 2, idummy, dummy, 8, nullptr) == DVM_ENOSPACE);                                       // workspace too small
    EXPECT(dvm_knn_neg_f32(dummy, dummy, 1, 4, 4, 4, 2, idummy, nullptr, 0, nullptr) == DVM_ENOSPACE);
    EXPECT(dvm_pair_fwd_f32(nullptr, nullptr, nullptr, nullptr, 1, 64, 64, -1.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                            nullptr, nullptr, nullptr, nullptr, nullptr, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                            nullptr, 0, nullptr) == DVM_EINVAL);
    EXPECT(dvm_rot6d_f32(nullptr, 4, nullptr, nullptr) == DVM_EINVAL);
    EXPECT(dvm_pair_geometry_f32(nullptr, nullptr, 1, 64, 64, nullptr, nullptr, 1, nullptr, 0, nullptr) == DVM_EINVAL);
    EXPECT(dvm_pair_geometry_f32(dummy, dummy, 1, 64, 64, idummy, idummy, 1, nullptr, 0, nullptr) == DVM_ENOSPACE);
    EXPECT(dvm_graph_geodesics_f64(nullptr, nullptr, 10, 4, nullptr, nullptr) == DVM_EINVAL);
    EXPECT(dvm_profile_read(nullptr, nullptr) == DVM_EINVAL);
    EXPECT(dvm_profile_enable(0) == DVM_EINVAL);
    // context registry: destroying with nothing registered, twice, is fine; set_overlap returns the previous value
    EXPECT(dvm_pair_destroy() == DVM_OK && dvm_pair_destroy() == DVM_OK);
    const int prev_ov = dvm_pair_set_overlap(0);
    EXPECT(dvm_pair_set_overlap(prev_ov) == 0);
    printf(fails ? "san_host: %d check(s) failed\n" : "san_host: ok\n", fails);
    return fails ? 1 : 0;
}
