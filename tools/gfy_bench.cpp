// Stand-alone C++ driver of the C ABI (no Python, no torch): random weights,
// a synthetic 60k-node / 300k-edge shard, per-kernel device times.
//   hipcc -O2 tools/gfy_bench.cpp -Iinclude -Lginfinity_amd/csrc -lgfy -o tools/gfy_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "gfy.h"
#ifdef GFY_STAMPS
extern "C" int gfy_debug_stamps(unsigned long long*, int);
extern "C" int gfy_debug_real(unsigned long long*);
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
#define GK(x) do { int s_ = (x); if (s_ != 0) { printf("gfy error %d: %s (line %d)\n", s_, gfy_last_error(), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
  setenv("GPU_MAX_HW_QUEUES", "8", 0);   // one hardware queue per stream in flight (see bench.py)
  const int64_t N = argc > 1 ? atoll(argv[1]) : 60000;
  const int steps = argc > 2 ? atoi(argv[2]) : 200;
  const bool no_edges = argc > 3 && argv[3][0] == 'n';   // MLP-only timing
  const int L = 4000;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  // weight pack
  size_t bytes = gfy_weight_pack_bytes(7, 128, 4, 10, 128);
  std::vector<char> pack(bytes);
  uint32_t hd[8] = {0x31594647u, 1, 7, 128, 4, 10, 128, 1};
  memcpy(pack.data(), hd, 32);
  float* w = (float*)(pack.data() + 32);
  size_t nf = (bytes - 32) / 4;
  for (size_t i = 0; i < nf; ++i) w[i] = 0.08f * nd(rng);
  // make running_var positive: simplest is abs()+0.5 everywhere it matters; use layout offsets
  {
    size_t off = 128 * 7 + 128;
    for (int l = 0; l < 4; ++l) {
      off += 1 + 128 * 10 + 128 + 256 * 128 + 256;  // eps, edge_lin, mlp0
      off += 256 * 3;                                // bn w,b,mean
      for (int c = 0; c < 256; ++c) w[off + c] = 0.5f + fabsf(w[off + c]) * 4.f;
      off += 256;
      off += 128 * 256 + 128 + 256;                  // mlp4, ln
    }
  }
  gfy_encoder* enc = nullptr;
  GK(gfy_encoder_create(pack.data(), bytes, GFY_F16, 0, &enc));
  if (getenv("GFY_BENCH_SEPARATE_HEAD"))   // the last layer launch is then a plain one
    GK(gfy_encoder_set_option(enc, GFY_OPT_SEPARATE_HEAD, 1));
  const int layer_kernel = getenv("GFY_BENCH_LAYER_KERNEL") ? atoi(getenv("GFY_BENCH_LAYER_KERNEL")) : -1;
  GK(gfy_encoder_set_option(enc, GFY_OPT_LAYER_KERNEL, layer_kernel));
  const int stagger = getenv("GFY_BENCH_STAGGER") ? atoi(getenv("GFY_BENCH_STAGGER")) : -1;
  GK(gfy_encoder_set_option(enc, GFY_OPT_STAGGER, stagger));
  const int prio = getenv("GFY_BENCH_PRIORITY") ? atoi(getenv("GFY_BENCH_PRIORITY")) : -1;
  GK(gfy_encoder_set_option(enc, GFY_OPT_PRIORITY, prio));
  // graph: records of L nodes: backbone both ways, skip2 both ways, random matching both ways
  const int64_t recs = N / L;
  std::vector<int32_t> src, dst; std::vector<uint8_t> typ;
  std::vector<int64_t> node_ptr(recs + 1), edge_ptr(recs + 1);   // record boundaries (gfy_shard, ABI 4)
  for (int64_t r = 0; r < recs; ++r) {
    node_ptr[r] = r * L; edge_ptr[r] = (int64_t)src.size();
    int32_t b = (int32_t)(r * L);
    for (int i = 0; i + 1 < L; ++i) { src.push_back(b + i); dst.push_back(b + i + 1); typ.push_back(0); }
    for (int i = 0; i + 1 < L; ++i) { src.push_back(b + i + 1); dst.push_back(b + i); typ.push_back(1); }
    std::vector<int32_t> perm(L); for (int i = 0; i < L; ++i) perm[i] = i;
    std::shuffle(perm.begin(), perm.end(), rng);
    for (int i = 0; i < L; i += 2) { src.push_back(b + perm[i]); dst.push_back(b + perm[i + 1]); typ.push_back(2); }
    for (int i = 0; i < L; i += 2) { src.push_back(b + perm[i + 1]); dst.push_back(b + perm[i]); typ.push_back(3); }
    for (int i = 0; i + 2 < L; ++i) { src.push_back(b + i); dst.push_back(b + i + 2); typ.push_back(4);
                                      src.push_back(b + i + 2); dst.push_back(b + i); typ.push_back(5); }
    for (int i = 0; i < 6; ++i) { src.push_back(b + rng() % L); dst.push_back(b + rng() % L); typ.push_back(2); }
  }
  node_ptr[recs] = N; edge_ptr[recs] = (int64_t)src.size();   // (N % L trailing nodes: isolated, in the last record)
  if (no_edges) { src.resize(1); dst.resize(1); typ.resize(1); }
  const int64_t E = (int64_t)src.size();
  // GFY_BENCH_RECORDS=0: the COO calls go without record boundaries (k_csr_count + k_encode_setup_coo)
  const bool with_records = !no_edges && !(getenv("GFY_BENCH_RECORDS") && atoi(getenv("GFY_BENCH_RECORDS")) == 0);
  std::vector<int32_t> ei(2 * E);
  memcpy(ei.data(), src.data(), E * 4); memcpy(ei.data() + E, dst.data(), E * 4);
  std::vector<float> x(N * 7);
  for (auto& v : x) v = nd(rng);
  float* dx; int32_t *dei, *drp, *dcol; uint8_t *det, *dtyp; void *dout, *ws1, *ws2;
  CK(hipMalloc(&dx, N * 7 * 4)); CK(hipMalloc(&dei, 2 * E * 4)); CK(hipMalloc(&det, E));
  CK(hipMalloc(&drp, (N + 1) * 4)); CK(hipMalloc(&dcol, E * 4)); CK(hipMalloc(&dtyp, E));
  CK(hipMalloc(&dout, N * 128 * 2));
  size_t b1 = gfy_csr_workspace_bytes(N, E), b2 = gfy_encode_workspace_bytes(enc, N, E);
  CK(hipMalloc(&ws1, b1)); CK(hipMalloc(&ws2, b2));
  CK(hipMemcpy(dx, x.data(), N * 7 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dei, ei.data(), 2 * E * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(det, typ.data(), E, hipMemcpyHostToDevice));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  for (int i = 0; i < 20; ++i) {
    GK(gfy_build_csr(dei, det, N, E, drp, dcol, dtyp, ws1, b1, s));
    GK(gfy_encode(enc, dx, drp, dcol, dtyp, N, E, nullptr, dout, GFY_F16, 1, ws2, b2, s));
  }
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < steps; ++i) {
    GK(gfy_build_csr(dei, det, N, E, drp, dcol, dtyp, ws1, b1, s));
    GK(gfy_encode(enc, dx, drp, dcol, dtyp, N, E, nullptr, dout, GFY_F16, 1, ws2, b2, s));
  }
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("N=%lld E=%lld  csr+encode: %.1f us/step  -> %.1f M nodes/s\n", (long long)N, (long long)E, 1e3 * ms / steps, N * steps / ms / 1e3);
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < steps; ++i)
    GK(gfy_encode(enc, dx, drp, dcol, dtyp, N, E, nullptr, dout, GFY_F16, 1, ws2, b2, s));
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("encode only: %.1f us/step\n", 1e3 * ms / steps);
  // the whole seam as one call (gfy_encode_coo_batch of one shard, with its record boundaries:
  // 1 + 4 launches; without them gfy_encode_coo's 2 + 4) on a workspace cleared once
  int64_t *dnp = nullptr, *dep = nullptr;
  CK(hipMalloc(&dnp, (recs + 1) * 8)); CK(hipMalloc(&dep, (recs + 1) * 8));
  CK(hipMemcpy(dnp, node_ptr.data(), (recs + 1) * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dep, edge_ptr.data(), (recs + 1) * 8, hipMemcpyHostToDevice));
  auto shard_of = [&](void* out) {
    gfy_shard one{};
    one.node_features = dx; one.edge_index = dei; one.edge_types = det; one.out = out;
    one.n_nodes = N; one.n_edges = E;
    if (with_records) { one.node_ptr = dnp; one.edge_ptr = dep; one.n_records = recs; }
    return one;
  };
  auto encode_coo = [&](gfy_encoder* which, void* out, void* ws, size_t bytes, hipStream_t on) {
    const gfy_shard one = shard_of(out);
    GK(gfy_encode_coo_batch(which, &one, 1, GFY_F16, 1, ws, bytes, on));
  };
  const gfy_shard probe = shard_of(dout);
  const size_t b3 = gfy_encode_coo_batch_workspace_bytes(enc, &probe, 1);
  void* ws3; CK(hipMalloc(&ws3, b3)); CK(hipMemsetAsync(ws3, 0, b3, s));
  for (int i = 0; i < 20; ++i) encode_coo(enc, dout, ws3, b3, s);
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < steps; ++i) encode_coo(enc, dout, ws3, b3, s);
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("gfy_encode_coo%s: %.1f us/step  -> %.1f M nodes/s\n", with_records ? " (record boundaries)" : "", 1e3 * ms / steps, N * steps / ms / 1e3);
  const int max_lanes = getenv("GFY_BENCH_STREAMS") ? atoi(getenv("GFY_BENCH_STREAMS")) : 4;
  for (int lanes = 2; lanes <= max_lanes; lanes += lanes < 4 ? 1 : 2) {  // independent shards in flight on several streams
    std::vector<gfy_encoder*> encs(lanes); std::vector<hipStream_t> ss(lanes);
    std::vector<void*> outs(lanes), wb(lanes);
    for (int q = 0; q < lanes; ++q) {
      GK(gfy_encoder_create(pack.data(), bytes, GFY_F16, 0, &encs[q])); CK(hipStreamCreateWithFlags(&ss[q], hipStreamNonBlocking));
      GK(gfy_encoder_set_option(encs[q], GFY_OPT_LAYER_KERNEL, layer_kernel));
      GK(gfy_encoder_set_option(encs[q], GFY_OPT_STAGGER, stagger));
      GK(gfy_encoder_set_option(encs[q], GFY_OPT_PRIORITY, prio));
      CK(hipMalloc(&outs[q], N * 128 * 2)); CK(hipMalloc(&wb[q], b3)); CK(hipMemset(wb[q], 0, b3));
    }
    auto step = [&](int i) {
      const int q = i % lanes;
      encode_coo(encs[q], outs[q], wb[q], b3, ss[q]);
    };
    const int many = steps * lanes;   // the same time in flight as the one-stream loop
    for (int i = 0; i < many / 2; ++i) step(i);
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < many; ++i) step(i);
    CK(hipDeviceSynchronize());
    double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
    printf("%d streams: %.1f us/step -> %.1f M nodes/s\n", lanes, us / many, N * many / us);
    for (int q = 0; q < lanes; ++q) gfy_encoder_destroy(encs[q]);
  }
  GK(gfy_encoder_set_timing(enc, 1));
  double sum[16] = {0}; int cnt = 0;
  for (int i = 0; i < 50; ++i) {
    GK(gfy_encode(enc, dx, drp, dcol, dtyp, N, E, nullptr, dout, GFY_F16, 1, ws2, b2, s));
    float t[16]; GK(gfy_encoder_get_timing(enc, t, 16, &cnt));
    for (int k = 0; k < cnt; ++k) sum[k] += t[k];
  }
  printf("per-kernel us:");
  for (int k = 0; k < cnt; ++k) printf(" %.1f", 1e3 * sum[k] / 50);
  printf("\n");
#ifdef GFY_STAMPS
  {
    static unsigned long long st[256][16];
    gfy_debug_stamps(nullptr, 1);
    const int reps = 20;
    for (int i = 0; i < reps; ++i)
      GK(gfy_encode(enc, dx, drp, dcol, dtyp, N, E, nullptr, dout, GFY_F16, 1, ws2, b2, s));
    CK(hipStreamSynchronize(s));
    gfy_debug_stamps(&st[0][0], 0);
    double sum[16] = {0};
    for (int b = 0; b < 256; ++b) for (int k = 0; k < 16; ++k) sum[k] += (double)st[b][k];
    if (layer_kernel == 5) {   // three 4-wave workgroups per CU: first, second and third residents apart
      const char* names[9] = {"prologue", "gather, channels 0..63", "second stage fill (issue + wait)", "gather, channels 64..127",
                              "c0 | c1 + barrier", "products (8 barriers, c2..c7)", "LayerNorm + store (head: LN)",
                              "wait for the next stage (head: normalise + store + wait)", "head launch: head pipeline"};
      for (int res = 0; res < 3; ++res) {
        double hs[16] = {0};
        for (int b = 64 * res; b < 64 * (res + 1); ++b) for (int k = 0; k < 16; ++k) hs[k] += (double)st[b][k];
        const double rounds = hs[11] > 0 ? hs[11] : 1;
        printf("three-workgroup layer kernel, resident %d of the CUs, shader cycles per round (wave 0):\n", res);
        for (int k = 1; k < 9; ++k) printf("  %-56s %8.0f\n", names[k], hs[k] / rounds);
        printf("  rounds per workgroup and launch %.2f, whole wave %.0f cycles per launch, %.0f per round\n",
               rounds / (64.0 * reps * 4), hs[10] / (64.0 * reps * 4), hs[10] / rounds);
      }
      static unsigned long long real[512][2];
      gfy_debug_real(&real[0][0]);
      double life = 0; int used = 0;
      for (int b = 0; b < 512; ++b)
        if (real[b][1] > real[b][0]) life += (real[b][1] - real[b][0]) / 100.0, ++used;
      if (used) printf("  last launch: workgroup lifetime %.1f us (mean of %d) -> %.2f GHz\n", life / used, used,
                       sum[10] / (192.0 * reps * 4) / (life / used) / 1e3);
    } else if (layer_kernel == 4 || (layer_kernel < 0 && N > 65536)) {   // windowed rounds (the default for several rounds of tiles): two 4-wave workgroups per CU, first and second residents apart
      const char* names[8] = {"prologue", "own rows + gather", "c0 | c1 + barrier", "products (4 barriers, c2, c3)",
                              "head launch: LN..head pipeline", "head launch: stage DMA issue", "LayerNorm + store (head: LN)", "wait for the next stage (head: normalise + store + wait)"};
      for (int half = 0; half < 2; ++half) {
        double hs[16] = {0};
        for (int b = 128 * half; b < 128 * (half + 1); ++b) for (int k = 0; k < 16; ++k) hs[k] += (double)st[b][k];
        const double rounds = hs[11] > 0 ? hs[11] : 1;
        printf("windowed layer kernel, %s workgroups of the CUs, shader cycles per round (wave %d):\n", half ? "SECOND" : "FIRST", 0);
        for (int k = 1; k < 8; ++k) printf("  %-56s %8.0f\n", names[k], hs[k] / rounds);
        printf("  inside the products, waits at step 12 / 28 / 44 / 60, stage DMA issue: %.0f %.0f %.0f %.0f %.0f\n",
               hs[8] / rounds, hs[9] / rounds, hs[12] / rounds, hs[13] / rounds, (hs[14] - hs[13]) / rounds);
        if (hs[15] > 0) printf("  diagnostic build: stage fill, last request -> landed: %.0f cycles\n", (hs[15] - hs[14]) / rounds);
        printf("  rounds per workgroup and launch %.2f, whole wave %.0f cycles per launch, %.0f per round\n",
               rounds / (128.0 * reps * 4), hs[10] / (128.0 * reps * 4), hs[10] / rounds);
      }
      static unsigned long long real[512][2];
      gfy_debug_real(&real[0][0]);
      double life = 0; int used = 0;
      for (int b = 0; b < 512; ++b)
        if (real[b][1] > real[b][0]) life += (real[b][1] - real[b][0]) / 100.0, ++used;
      if (used) printf("  last launch: workgroup lifetime %.1f us (mean of %d) -> %.2f GHz\n", life / used, used,
                       sum[10] / (256.0 * reps * 4) / (life / used) / 1e3);
    } else if (layer_kernel >= 3 || (layer_kernel < 0 && N > 65536)) {
      const double rounds = sum[11] > 0 ? sum[11] : 1;
      const char* names[8] = {"prologue (image, first plan head + stage)", "own rows + gather",
                              "[W0 | W1] over the stages + barrier", "update MLP (both products)",
                              "next plan head requested + barrier (weights spent)",
                              "next plan head read + stage DMA issue + slot words",
                              "LayerNorm + residual + store", "wait for the next stage"};
      printf("persistent-rounds layer kernel, shader cycles (stamped wave of workgroups 0..255):\n");
      printf("  %-56s %8.0f per launch\n", names[0], sum[0] / (256.0 * reps * 4));
      for (int k = 1; k < 8; ++k) printf("  %-56s %8.0f per round\n", names[k], sum[k] / rounds);
      printf("  rounds per workgroup and launch %.2f, whole wave %.0f cycles per launch, %.0f per round\n",
             rounds / (256.0 * reps * 4), sum[10] / (256.0 * reps * 4), sum[10] / rounds);
      {   // the clock the chip held: shader cycles of the stamped wave / its workgroup's lifetime
        static unsigned long long real[512][2];
        gfy_debug_real(&real[0][0]);
        double life = 0; int used = 0;
        for (int b = 0; b < 256; ++b)
          if (real[b][1] > real[b][0]) life += (real[b][1] - real[b][0]) / 100.0, ++used;
        if (used)
          printf("  last launch: workgroup lifetime %.1f us (mean of %d) -> %.2f GHz shader clock while it ran\n",
                 life / used, used, sum[10] / (256.0 * reps * 4) / (life / used) / 1e3);
      }
    } else {
      const double plain = sum[11], headed = sum[13], all = plain + headed;
      const char* names[10] = {"plan + own rows (hop 1)", "far rows DMA + wait (hop 2)", "barrier 1",
                               "gather", "[W0 | W1] over own stage + barrier 2",
                               "update MLP (both products, one pipeline)", "head: barrier 4 (image resident)", "head: both products", "LayerNorm + store",
                               "head: normalise + store"};
      printf("layer kernel phases, shader cycles per launch (one wave of each workgroup — wave 0 unless the\n"
             "library was built with -DGFY_STAMP_WAVE=n —, mean):\n");
      for (int k = 0; k < 9; ++k)
        if (k == 6 || k == 7) printf("  %-42s %8.0f (last launch only)\n", names[k], sum[k] / headed);
        else printf("  %-42s %8.0f\n", names[k], sum[k] / all);
      printf("  %-42s %8.0f (last launch only)\n", names[9], sum[9] / headed);
      printf("  whole wave: %.0f cycles (plain layer), %.0f (with head)\n", sum[10] / plain, sum[12] / headed);
      static unsigned long long real[512][2];
      gfy_debug_real(&real[0][0]);
      unsigned long long b0 = ~0ull, b1 = 0, e0 = ~0ull, e1 = 0;
      int used = 0;
      for (int b = 0; b < 512; ++b) {
        if (!real[b][1]) continue;
        ++used;
        b0 = real[b][0] < b0 ? real[b][0] : b0; b1 = real[b][0] > b1 ? real[b][0] : b1;
        e0 = real[b][1] < e0 ? real[b][1] : e0; e1 = real[b][1] > e1 ? real[b][1] : e1;
      }
      if (used && getenv("GFY_BENCH_WORKGROUPS")) {   // lifetimes per XCD, and the slowest few
        double per_xcd[8] = {0}, worst_xcd[8] = {0}; int in_xcd[8] = {0};
        for (int b = 0; b < 512; ++b) {
          if (!real[b][1]) continue;
          const double us = (real[b][1] - real[b][0]) / 100.0;
          per_xcd[b & 7] += us; in_xcd[b & 7]++;
          worst_xcd[b & 7] = us > worst_xcd[b & 7] ? us : worst_xcd[b & 7];
        }
        for (int x = 0; x < 8; ++x)
          printf("  XCD %d: %d workgroups, lifetime mean %.2f us, longest %.2f us\n", x, in_xcd[x],
                 per_xcd[x] / (in_xcd[x] ? in_xcd[x] : 1), worst_xcd[x]);
        printf("  workgroup: start, end (us after the first start)\n");
        for (int b = 0; b < 512; ++b)
          if (real[b][1]) printf("  %3d %6.2f %6.2f\n", b, (real[b][0] - b0) / 100.0, (real[b][1] - b0) / 100.0);
      }
      if (used)
        printf("last layer launch, real time (100 MHz): %d workgroups, first start -> last start %.2f us, "
               "first end %.2f us, last end %.2f us after the first start\n",
               used, (b1 - b0) / 100.0, (e0 - b0) / 100.0, (e1 - b0) / 100.0);
    }
  }
#endif
  gfy_encoder_destroy(enc);
  return 0;
}
