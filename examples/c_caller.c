/*
 * A caller with nothing but include/somhip.h: the boundary of the hot path is a C ABI, this is C.
 *
 *   gcc -O2 -I include examples/c_caller.c -o c_caller -L xpysom_dask_amd -lsomhip -Wl,-rpath,$PWD/xpysom_dask_amd -lm
 *   ./c_caller rows.f32 weights.f32 X Y D N schedule.f64 out_weights.f32 out_bmu.i32
 *
 * Reads N x D float32 rows, an X*Y x D float32 codebook and a schedule of (sigma, learning rate) float64 pairs, one per
 * epoch (the host keeps the decay functions, as in the reference: decays.py), trains those epochs (euclidean,
 * gaussian, rectangular, float32 precision, float64 neighbourhood as with the reference's default exponential decay),
 * writes the trained codebook and the BMU of every row, prints the quantization error.
 * tests/test_c_abi.py checks the two files against the same run through the Python host.
 */
#include <stdio.h>
#include <stdlib.h>

#include "somhip.h"

static void* slurp(const char* path, size_t bytes) {
    FILE* f = fopen(path, "rb");
    void* p = malloc(bytes ? bytes : 1);
    if (!f || !p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot read %zu bytes of %s\n", bytes, path); exit(2); }
    fclose(f);
    return p;
}
static void dump(const char* path, const void* p, size_t bytes) {
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
    fclose(f);
}
static long file_bytes(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fclose(f);
    return n;
}

#define CHECK(h, call)                                                                    \
    do {                                                                                  \
        if ((call) != 0) { fprintf(stderr, "%s: %s\n", #call, som_last_error(h)); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc != 10) { fprintf(stderr, "usage: %s rows weights X Y D N schedule out_weights out_bmu\n", argv[0]); return 2; }
    const int X = atoi(argv[3]), Y = atoi(argv[4]), D = atoi(argv[5]);
    const long N = atol(argv[6]);
    const int T = (int)(file_bytes(argv[7]) / (2 * sizeof(double)));
    double* sched = slurp(argv[7], (size_t)T * 2 * sizeof(double));
    float* rows = slurp(argv[1], (size_t)N * D * sizeof(float));
    float* w = slurp(argv[2], (size_t)X * Y * D * sizeof(float));

    som_config cfg = {0};
    cfg.x = X; cfg.y = Y; cfg.input_len = D;
    cfg.distance = SOM_DIST_EUCLIDEAN; cfg.neighborhood = SOM_NEIGH_GAUSSIAN; cfg.topology = SOM_TOPO_RECTANGULAR;
    cfg.precision = SOM_PREC_EXACT; cfg.device = 0; cfg.std_coeff = 0.5;
    som_handle* h = NULL;
    if (som_create(&cfg, &h) != 0) { fprintf(stderr, "som_create: %s\n", som_last_error(NULL)); return 3; }
    printf("%s, %d device(s)\n", som_version(), som_device_count());
    CHECK(h, som_set_weights(h, w));
    CHECK(h, som_set_data(h, rows, N));
    for (int t = 0; t < T; ++t)   /* accumulate (+ all-reduce, had som_comm_init been called) + merge */
        CHECK(h, som_epoch(h, sched[2 * t], sched[2 * t + 1], 1));
    CHECK(h, som_get_weights(h, w));
    int32_t* bmu = malloc((size_t)(N ? N : 1) * sizeof(int32_t));
    CHECK(h, som_bmu(h, rows, N, SOM_BMU_ACTIVATION, bmu));
    double qe = 0.0;
    CHECK(h, som_quantization_error(h, rows, N, &qe));
    printf("quantization error %.6f\n", qe);
    dump(argv[8], w, (size_t)X * Y * D * sizeof(float));
    dump(argv[9], bmu, (size_t)N * sizeof(int32_t));
    som_destroy(h);
    free(rows); free(w); free(bmu); free(sched);
    return 0;
}
