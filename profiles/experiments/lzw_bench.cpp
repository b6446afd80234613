// lzw_bench.cpp -- encoder rate of csrc/oip_tiff.hpp on sensor-like 4-sample data: OIP_TIFF_THREADS=N ./lzw_bench (build: hipcc -O3 -std=c++17 -pthread -I ../../opticalimageprocessor_amd/csrc)
#include "oip_tiff.hpp"
#include <chrono>
#include <random>
int main(){
  const int W=7500; const long H=getenv("LZW_H") ? atol(getenv("LZW_H")) : 4000; const int S=4;
  std::vector<uint16_t> d((size_t)W*H*S);
  std::mt19937 rng(1); std::normal_distribution<float> g(1800.f,300.f);
  for(auto&v:d){ float x=g(rng); v=(uint16_t)(x<64?64:x>4095?4095:x);}
  auto t0=std::chrono::steady_clock::now();
  OIPGPU::write_tiff_u16((getenv("LZW_OUT") ? getenv("LZW_OUT") : "/dev/shm/lzwbench.tiff"), d.data(), W, H, S, false, OIPGPU::TIFF_LZW);
  double s=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
  printf("%.1f MB in %.2f s = %.1f MB/s (threads %d)\n", d.size()*2/1e6, s, d.size()*2/1e6/s, OIPGPU::tiffdetail::worker_count());
}
