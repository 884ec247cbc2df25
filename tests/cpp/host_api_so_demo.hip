// Exercises the generated C++ host API of a serial-chain robot for T = float and T = double: forward_dynamics_gradient<T>, forward_dynamics<T>,
// idsva_so_host<T, true> (second-order inverse-dynamics derivatives at the forward-dynamics solution) and fdsva_so<T>.
// usage: host_api_so_demo <in.bin (N x 3n doubles)> <N> <out prefix> [threads of the second-order launches for T = double, 0 = suggested]
//        -> <prefix>_{f32,f64}_{dfdu,so,df2}.bin (doubles)
#include "grid.cuh"
#include <string>
#include <vector>

template <typename T>
static void dump(const std::string &path, const T *p, size_t count) {
    std::vector<double> out(count);
    for (size_t i = 0; i < count; i++) out[i] = static_cast<double>(p[i]);
    FILE *f = fopen(path.c_str(), "wb");
    fwrite(out.data(), sizeof(double), count, f);
    fclose(f);
}

template <typename T>
static void run(const std::vector<double> &in, int N, const std::string &prefix, int so_threads) {
    using namespace grid;
    const int n = NUM_JOINTS;
    robotModel<T> *d_robotModel = init_robotModel<T>();
    hipStream_t *streams = init_grid<T>();
    gridData<T> *hd_data = init_gridData<T>(N);
    for (size_t i = 0; i < in.size(); i++) hd_data->h_q_qd_u[i] = static_cast<T>(in[i]);
    const T g = static_cast<T>(9.81);
    const int gpb = SUGGESTED_THREADS / GRID_LANES_PER_SOLVE;
    dim3 blocks((N + gpb - 1) / gpb, 1, 1), threads(SUGGESTED_THREADS, 1, 1);
    forward_dynamics_gradient<T>(hd_data, d_robotModel, g, N, blocks, threads, streams);
    dump(prefix + "_dfdu.bin", hd_data->h_df_du, (size_t)N * 2 * n * n);
    forward_dynamics<T>(hd_data, d_robotModel, g, N, blocks, threads, streams);  // h_qdd <- FD(q, qd, u): the point the second-order tensors are taken at
    // (the suggested block sizes are sized for T = float: a 12-joint robot's 4 n^3 record in double precision needs fewer solves per block)
    const int thr_so = so_threads > 0 ? so_threads : IDSVA_SO_SUGGESTED_THREADS;
    const int gpb_so = thr_so / GRID_LANES_PER_SOLVE;
    idsva_so_host<T, true>(hd_data, d_robotModel, g, N, dim3((N + gpb_so - 1) / gpb_so, 1, 1), dim3(thr_so, 1, 1), streams);
    dump(prefix + "_so.bin", hd_data->h_idsva_so, (size_t)N * 4 * n * n * n);
    {   // the header's general SUGGESTED_THREADS launch dims must work for every kernel (the reference's hosts accept any thread_dimms):
        // lane groups beyond the kernel's own cap retire and the wrapper sizes the LDS for the capped count
        std::vector<T> first(hd_data->h_idsva_so, hd_data->h_idsva_so + (size_t)N * 4 * n * n * n);
        for (size_t i = 0; i < first.size(); i++) hd_data->h_idsva_so[i] = static_cast<T>(-1);
        if (so_threads > 0) { idsva_so_host<T, true>(hd_data, d_robotModel, g, N, dim3(N, 1, 1), dim3(so_threads, 1, 1), streams); }  // (another grid shape instead)
        else { idsva_so_host<T, true>(hd_data, d_robotModel, g, N, blocks, threads, streams); }
        size_t bad = 0;
        for (size_t i = 0; i < first.size(); i++) bad += (first[i] != hd_data->h_idsva_so[i]);
        printf("idsva_so with SUGGESTED_THREADS: mismatches = %zu\n", bad);
    }
    const int thr_fd = so_threads > 0 ? so_threads : FDSVA_SO_SUGGESTED_THREADS;
    const int gpb_fd = thr_fd / GRID_LANES_PER_SOLVE;
    fdsva_so<T>(hd_data, d_robotModel, g, N, dim3((N + gpb_fd - 1) / gpb_fd, 1, 1), dim3(thr_fd, 1, 1), streams);
    dump(prefix + "_df2.bin", hd_data->h_df2, (size_t)N * 4 * n * n * n);
    {
        std::vector<T> first(hd_data->h_df2, hd_data->h_df2 + (size_t)N * 4 * n * n * n);
        for (size_t i = 0; i < first.size(); i++) hd_data->h_df2[i] = static_cast<T>(-1);
        if (so_threads > 0) { fdsva_so<T>(hd_data, d_robotModel, g, N, dim3(N, 1, 1), dim3(so_threads, 1, 1), streams); }
        else { fdsva_so<T>(hd_data, d_robotModel, g, N, blocks, threads, streams); }
        size_t bad = 0;
        for (size_t i = 0; i < first.size(); i++) bad += (first[i] != hd_data->h_df2[i]);
        printf("fdsva_so with SUGGESTED_THREADS: mismatches = %zu\n", bad);
    }
    close_grid<T>(streams, d_robotModel, hd_data);
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage\n"); return 2; }
    const int N = atoi(argv[2]);
    std::vector<double> in((size_t)N * 3 * grid::NUM_JOINTS);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(in.data(), sizeof(double), in.size(), f) != in.size()) { fprintf(stderr, "bad input\n"); return 2; }
    fclose(f);
    const int so_threads_f64 = argc > 4 ? atoi(argv[4]) : 0;
    run<float>(in, N, std::string(argv[3]) + "_f32", 0);
    run<double>(in, N, std::string(argv[3]) + "_f64", so_threads_f64);
    printf("done\n");
    return 0;
}
