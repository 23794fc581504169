#include "../../tscd_amd/csrc_host/lap.cpp"
#include <chrono>
#include <cstdio>
int main(int argc, char** argv) {
  const int K = 8, n = 500;
  std::vector<uint16_t> mats((size_t)K * n * n);
  std::vector<int64_t> cols((size_t)K * n);
  FILE* f = fopen("tools/lap_ab/mats.bin", "rb"); fread(mats.data(), 2, mats.size(), f); fclose(f);
  f = fopen("tools/lap_ab/cols.bin", "rb"); fread(cols.data(), 8, cols.size(), f); fclose(f);
  std::vector<int64_t> r(n), c(n);
  int reps = argc > 1 ? atoi(argv[1]) : 5;
  double best = 1e9;
  for (int rep = 0; rep < reps; ++rep) {
    tscd_lap_debug_steps(1);
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < K; ++k) {
      int st = tscd_lap_solve_f16(mats.data() + (size_t)k * n * n, n, nullptr, n, n, r.data(), c.data());
      if (st != 0) { printf("status %d\n", st); return 1; }
      for (int i = 0; i < n; ++i) if (c[i] != cols[(size_t)k * n + i]) { printf("MISMATCH matrix %d row %d\n", k, i); return 1; }
    }
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    long steps = tscd_lap_debug_steps(1);
    if (dt < best) best = dt;
    printf("rep %d: %.3f ms / matrix, %ld passes / matrix, %.1f ns / pass\n", rep, dt / K * 1e3, steps / K, dt / steps * 1e9);
  }
  return 0;
}
