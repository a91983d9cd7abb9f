/* A C client of the drop-in boundary (include/savad.h) -- what a maintainer's cgo / JNI / plain-C binding would link
 * against: compiled with gcc as C99 by tests/test_abi_and_host.py (no GPU needed to build and link it) and run on the
 * GPU box by tests/test_gpu_parity.py::test_c_client_of_the_abi.  No HIP headers: device memory comes from
 * hipMalloc / hipMemcpy looked up in libamdhip64 at run time, the way a foreign-language host would.
 * usage: abi_client <path to libamdhip64.so>   -> prints "ok <checksum>" */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "savad.h"

typedef int (*malloc_fn)(void**, size_t);
typedef int (*memcpy_fn)(void*, const void*, size_t, int);
typedef int (*sync_fn)(void);

int main(int argc, char** argv) {
    void* hip = dlopen(argc > 1 ? argv[1] : "libamdhip64.so", RTLD_NOW);
    if (!hip) return 2;
    malloc_fn hipMalloc = (malloc_fn)dlsym(hip, "hipMalloc");
    memcpy_fn hipMemcpy = (memcpy_fn)dlsym(hip, "hipMemcpy");
    sync_fn hipDeviceSynchronize = (sync_fn)dlsym(hip, "hipDeviceSynchronize");
    if (!hipMalloc || !hipMemcpy || !hipDeviceSynchronize) return 2;

    const int F = 80, L = 3, B = 3, T = 7;
    savad_config cfg = {F, L, 128};
    savad_handle h = NULL;
    if (savad_create(&cfg, &h) != SAVAD_OK) { fprintf(stderr, "%s\n", savad_last_error()); return 1; }
    /* every parameter: a deterministic pattern (LayerNorm weights near one) */
    const int np = savad_num_params(h);
    for (int i = 0; i < np; ++i) {
        const char* key = savad_param_key(h, i);
        const size_t n = savad_param_numel(h, i);
        float* w = (float*)malloc(n * sizeof(float));
        const int is_ln_w = strstr(key, "layer_norm.weight") != NULL;
        for (size_t j = 0; j < n; ++j) w[j] = (is_ln_w ? 1.0f : 0.0f) + 0.05f * sinf(0.37f * (float)j + (float)i);
        if (savad_set_param(h, key, w, n, NULL) != SAVAD_OK) { fprintf(stderr, "%s\n", savad_last_error()); return 1; }
        free(w);
    }
    size_t ws_bytes = 0;
    if (savad_workspace_bytes(h, B, T, &ws_bytes) != SAVAD_OK) return 1;
    void *dx = NULL, *dout = NULL, *dws = NULL;
    if (hipMalloc(&dx, sizeof(float) * B * T * F) || hipMalloc(&dout, sizeof(float) * B * T * 2) || hipMalloc(&dws, ws_bytes ? ws_bytes : 16)) return 2;
    float* x = (float*)malloc(sizeof(float) * B * T * F);
    for (int i = 0; i < B * T * F; ++i) x[i] = -5.0f + 4.0f * cosf(0.11f * (float)i);
    if (hipMemcpy(dx, x, sizeof(float) * B * T * F, 1 /* host to device */)) return 2;
    if (savad_forward(h, (const float*)dx, B, T, (float*)dout, dws, ws_bytes, NULL) != SAVAD_OK) { fprintf(stderr, "%s\n", savad_last_error()); return 1; }
    hipDeviceSynchronize();
    float out[3 * 7 * 2];
    if (hipMemcpy(out, dout, sizeof(out), 2 /* device to host */)) return 2;
    double sum = 0.0;
    for (int i = 0; i < B * T; ++i) {
        const double p = exp((double)out[2 * i]) + exp((double)out[2 * i + 1]);  /* log-softmax: the two probabilities sum to one */
        if (!(fabs(p - 1.0) < 1e-5)) { fprintf(stderr, "row %d: p0 + p1 = %.9f\n", i, p); return 1; }
        sum += out[2 * i];
    }
    /* error behaviour is part of the boundary */
    if (savad_forward(h, (const float*)dx, B, T, (float*)dout, dws, 0, NULL) != SAVAD_E_INVALID) return 1;
    if (savad_set_param(h, "no.such.key", x, 1, NULL) != SAVAD_E_NOKEY) return 1;
    savad_destroy(h);
    printf("ok %.6f\n", sum);
    return 0;
}
