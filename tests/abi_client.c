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
    /* round 5: bf16 operands through the same entry point -- at T <= 32 the whole forward is ONE launch (csrc/savad_packed_bf16.h) */
    {
        if (savad_set_precision(h, 1) != SAVAD_OK || savad_workspace_bytes(h, B, T, &ws_bytes) != SAVAD_OK) return 1;
        void* dws16 = NULL;
        if (hipMalloc(&dws16, ws_bytes ? ws_bytes : 16)) return 2;
        if (savad_forward(h, (const float*)dx, B, T, (float*)dout, dws16, ws_bytes, NULL) != SAVAD_OK) { fprintf(stderr, "%s\n", savad_last_error()); return 1; }
        hipDeviceSynchronize();
        float out16[3 * 7 * 2];
        if (hipMemcpy(out16, dout, sizeof(out16), 2)) return 2;
        for (int i = 0; i < B * T * 2; ++i)
            if (!(fabs((double)out16[i] - (double)out[i]) < 2e-2)) { fprintf(stderr, "bf16 log-prob %d: %.6f vs %.6f\n", i, out16[i], out[i]); return 1; }
        if (savad_set_precision(h, 0) != SAVAD_OK) return 1;
    }
    /* round 5: the log-mel front-end, whole signal == its frame spans computed from slices of the audio (a rank's share of a sharded run) */
    {
        const int n = 16000 + 77, nf = savad_logmel_frames(n);
        float* y = (float*)malloc(sizeof(float) * n);
        for (int i = 0; i < n; ++i) y[i] = 0.3f * sinf(0.173f * (float)i) + 0.01f * cosf(1.9f * (float)i);
        void *dy = NULL, *dfeat = NULL, *dspan = NULL, *dmw = NULL;
        const size_t mw = savad_logmel_workspace_bytes(n) > savad_logmel_span_workspace_bytes(nf) ? savad_logmel_workspace_bytes(n) : savad_logmel_span_workspace_bytes(nf);
        if (hipMalloc(&dy, sizeof(float) * n) || hipMalloc(&dfeat, sizeof(float) * nf * 80) || hipMalloc(&dspan, sizeof(float) * nf * 80) || hipMalloc(&dmw, mw)) return 2;
        if (hipMemcpy(dy, y, sizeof(float) * n, 1)) return 2;
        if (savad_logmel((const float*)dy, n, (float*)dmw, (float*)dfeat, NULL) != SAVAD_OK) { fprintf(stderr, "%s\n", savad_last_error()); return 1; }
        const int cut = nf / 2;
        for (int part = 0; part < 2; ++part) {
            const int f0 = part ? cut : 0, fc = part ? nf - cut : cut;
            long first = 0, count = 0;
            if (savad_logmel_span_samples(n, f0, fc, &first, &count) != SAVAD_OK || first % 4 || first + count > n) return 1;
            if (savad_logmel_span((const float*)dy + first, first, count, n, f0, fc, (float*)dmw, (float*)dspan + (size_t)f0 * 80, NULL) != SAVAD_OK) {
                fprintf(stderr, "%s\n", savad_last_error());
                return 1;
            }
        }
        hipDeviceSynchronize();
        float* a = (float*)malloc(sizeof(float) * nf * 80);
        float* b = (float*)malloc(sizeof(float) * nf * 80);
        if (hipMemcpy(a, dfeat, sizeof(float) * nf * 80, 2) || hipMemcpy(b, dspan, sizeof(float) * nf * 80, 2)) return 2;
        if (memcmp(a, b, sizeof(float) * nf * 80) != 0) { fprintf(stderr, "log-mel spans differ from the whole signal's rows\n"); return 1; }
        if (savad_logmel_span((const float*)dy, 0, 100, n, 0, nf, (float*)dmw, (float*)dspan, NULL) != SAVAD_E_INVALID) return 1;  /* a slice that is too short */
        free(a); free(b); free(y);
    }
    savad_destroy(h);
    printf("ok %.6f\n", sum);
    return 0;
}
