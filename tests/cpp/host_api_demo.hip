// Exercises the GENERATED C++ host API exactly as a downstream GRiD/MPC program would (reference interface doc block,
// GRiDCodeGenerator.py:312-380): init_robotModel / init_grid / init_gridData / forward_dynamics_gradient<T> / close_grid,
// for T = float and T = double.  usage: host_api_demo <in.bin (N x 3n doubles)> <N> <out_f32.bin> <out_f64.bin> [threads per block]
#include "grid.cuh"
#include <vector>

template <typename T>
static void run(const std::vector<double> &in, int N, const char *out_path, int threads) {
    using namespace grid;
    robotModel<T> *d_robotModel = init_robotModel<T>();
    hipStream_t *streams = init_grid<T>();
    gridData<T> *hd_data = init_gridData<T>(N);
    for (size_t i = 0; i < in.size(); i++) hd_data->h_q_qd_u[i] = static_cast<T>(in[i]);
    if (threads <= 0) threads = SUGGESTED_THREADS;
    const int gpb = threads / GRID_LANES_PER_SOLVE;
    dim3 block_dimms((N + gpb - 1) / gpb, 1, 1), thread_dimms(threads, 1, 1);
    forward_dynamics_gradient<T>(hd_data, d_robotModel, static_cast<T>(9.81), N, block_dimms, thread_dimms, streams);
    std::vector<double> out((size_t)N * 2 * NUM_JOINTS * NUM_JOINTS);
    for (size_t i = 0; i < out.size(); i++) out[i] = static_cast<double>(hd_data->h_df_du[i]);
    // the (q,qd,qdd,Minv)-input overload fed by the stand-alone kernels must reproduce the same result
    forward_dynamics<T>(hd_data, d_robotModel, static_cast<T>(9.81), N, block_dimms, thread_dimms, streams);
    direct_minv<T>(hd_data, d_robotModel, N, block_dimms, thread_dimms, streams);
    forward_dynamics_gradient<T, true>(hd_data, d_robotModel, static_cast<T>(9.81), N, block_dimms, thread_dimms, streams);
    double worst = 0, scale = 0;
    for (size_t i = 0; i < out.size(); i++) {
        double d = out[i] - static_cast<double>(hd_data->h_df_du[i]);
        worst = d < 0 ? (-d > worst ? -d : worst) : (d > worst ? d : worst);
        double a = out[i] < 0 ? -out[i] : out[i];
        scale = a > scale ? a : scale;
    }
    printf("%s: overload consistency max|delta|/max|x| = %.3e\n", sizeof(T) == 4 ? "float" : "double", worst / scale);
    FILE *f = fopen(out_path, "wb");
    fwrite(out.data(), sizeof(double), out.size(), f);
    fclose(f);
    close_grid<T>(streams, d_robotModel, hd_data);
}

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage\n"); return 2; }
    const int N = atoi(argv[2]);
    std::vector<double> in((size_t)N * 3 * grid::NUM_JOINTS);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(in.data(), sizeof(double), in.size(), f) != in.size()) { fprintf(stderr, "bad input\n"); return 2; }
    fclose(f);
    const int threads = argc > 5 ? atoi(argv[5]) : 0;
    run<float>(in, N, argv[3], threads);
    run<double>(in, N, argv[4], threads);
    return 0;
}
