// Cost of itm_depth_stager in isolation: frames of 640x480 shorts from pinned host memory, one (or N) ahead, each converted by
// itm_update_view and followed by a spin kernel of ~80 us standing in for the fused frame.
// build: hipcc --offload-arch=gfx950 -O2 -I include tools/microbench/stager_bench.hip -L infinitam_amd -litmhip -Wl,-rpath,$PWD/infinitam_amd -o /tmp/stager_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <thread>
#include <atomic>
#include "itm_hip.h"

__global__ void spin_kernel(long long cycles, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 9999) *sink = 1;
}

int main(int argc, char** argv) {
  const int W = 640, H = 480, frames = 400;
  const int ahead = argc > 1 ? atoi(argv[1]) : 1;
  const int mode = argc > 2 ? atoi(argv[2]) : 0;          // 0 stager, 1 copy on the frame's stream, 2 stager fed by a thread of its own
  const long long spin = (argc > 3 ? atoll(argv[3]) : 80) * 100;   // wall clock: 100 MHz
  std::vector<int16_t*> host(8);
  for (auto& p : host) { if (hipHostMalloc((void**)&p, W * H * 2) != hipSuccess) return 1; for (int i = 0; i < W * H; ++i) p[i] = (int16_t)(1000 + i % 500); }
  hipStream_t st; hipStreamCreate(&st);
  float *depth, *scratch; hipMalloc((void**)&depth, W * H * 4); hipMalloc((void**)&scratch, W * H * 4);
  int16_t* slot; hipMalloc((void**)&slot, W * H * 2);
  const float intr[4] = {525, 525, 320, 240};
  itm_depth_stager* g = nullptr;
  if (itm_depth_stager_create(W, H, ahead + 2, &g)) { printf("create failed: %s\n", itm_last_error()); return 1; }
  for (int rep = 0; rep < 2; ++rep) {
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    int staged = -1;
    double hostUpload = 0;
    std::atomic<int> uploadedFrames{0};
    std::thread producer;
    if (mode == 2) producer = std::thread([&]() {
      for (int f = 0; f < frames; ++f) {
        while (itm_depth_stager_upload(g, host[f % 8]) != 0) std::this_thread::yield();      // (refused while every slot is taken)
        uploadedFrames.store(f + 1, std::memory_order_release);
      }
    });
    for (int k = 0; k < frames; ++k) {
      const int16_t* dev = slot;
      if (mode == 2) {
        while (uploadedFrames.load(std::memory_order_acquire) <= k) std::this_thread::yield();
        if (itm_depth_stager_acquire(g, (itm_stream)st, &dev)) { printf("acquire: %s\n", itm_last_error()); return 1; }
      } else if (mode == 0) {
        const auto h0 = std::chrono::steady_clock::now();
        while (staged < k + ahead && staged < frames - 1) { ++staged; if (itm_depth_stager_upload(g, host[staged % 8])) { printf("upload: %s\n", itm_last_error()); return 1; } }
        hostUpload += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
        if (itm_depth_stager_acquire(g, (itm_stream)st, &dev)) { printf("acquire: %s\n", itm_last_error()); return 1; }
      } else {
        hipMemcpyAsync(slot, host[k % 8], W * H * 2, hipMemcpyHostToDevice, st);
      }
      itm_update_view(dev, W, H, 1, 0.001f, 0.0f, intr, 0, 0, depth, scratch, nullptr, nullptr, (itm_stream)st);
      if (mode != 1) itm_depth_stager_release(g, (itm_stream)st);
      spin_kernel<<<1, 64, 0, st>>>(spin, nullptr);
    }
    hipStreamSynchronize(st);
    if (producer.joinable()) producer.join();
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("{\"mode\": \"%s\", \"ahead\": %d, \"us_per_frame\": %.1f, \"host_us_in_upload_calls\": %.1f}\n", mode == 1 ? "same stream" : mode == 2 ? "stager, producer thread" : "stager", ahead, us / frames, hostUpload / frames);
  }
  itm_depth_stager_destroy(g);
  return 0;
}
