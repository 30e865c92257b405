#include "../2022-entries_amd/csrc/host_fold64.hpp"
#include <chrono>
#include <cstdio>
using namespace msm;
int main() {
  Fp64 f{}; f.init<Bls12_377_Fq>();
  F64 a = f.one, b = f.from28;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 2000000; i++) f.mul(a, a, b);
  auto t1 = std::chrono::steady_clock::now();
  printf("mul %.1f ns (%llx)\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / 2e6, (unsigned long long)a.l[0]);
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 2000000; i++) f.add(a, a, b);
  t1 = std::chrono::steady_clock::now();
  printf("add %.1f ns (%llx)\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / 2e6, (unsigned long long)a.l[0]);
  Xyzz64 p{}; p.y = f.one; p.zz = f.one; p.x = b; p.zzz = a;
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 200000; i++) te64_dbl(f, p);
  t1 = std::chrono::steady_clock::now();
  printf("te64_dbl %.1f ns (%llx)\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / 2e5, (unsigned long long)p.x.l[0]);
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 2000; i++) f.invert(a, a);
  t1 = std::chrono::steady_clock::now();
  printf("invert %.1f us (%llx)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 2e3, (unsigned long long)a.l[0]);
}
