// How does v_mfma_f32_32x32x16_bf16 round?  One wave, hand-made operands; prints D[0][0] as hex for a list of cases
// next to what round-to-nearest-even / truncation of an exact sum would give.  (hipcc --offload-arch=gfx950 -O2)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A: [32 rows][16 k], B: [32 cols][16 k] (both as float, converted exactly to bf16), C: scalar broadcast
__global__ void probe(const float *A, const float *B, const float *C, float *D, int ncase) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  for (int c = 0; c < ncase; ++c) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
      a[j] = (__bf16)A[(c * 32 + r) * 16 + 8 * h + j];
      b[j] = (__bf16)B[(c * 32 + r) * 16 + 8 * h + j];
    }
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = C[c];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (lane == 0) D[c] = acc[0];   // D[row 0][col 0]
  }
}
// same with the fp32 MFMA (two k per instruction): k-ordered fma chain reference
__global__ void probe32(const float *A, const float *B, const float *C, float *D, int ncase) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  for (int c = 0; c < ncase; ++c) {
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = C[c];
    for (int kk = 0; kk < 8; ++kk) {
      const float a = A[(c * 32 + r) * 16 + 2 * kk + h], b = B[(c * 32 + r) * 16 + 2 * kk + h];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (lane == 0) D[c] = acc[0];
  }
}

struct Case { const char *name; float c; float a[16]; float b[16]; };
static unsigned bits(float x) { unsigned u; memcpy(&u, &x, 4); return u; }

int main() {
  const float u = ldexpf(1.f, -23);  // ulp(1.0)
  std::vector<Case> cs;
  auto add = [&](const char *n, float c, std::vector<std::pair<float, float>> prods) {
    Case k{}; k.name = n; k.c = c;
    for (size_t i = 0; i < prods.size() && i < 16; ++i) { k.a[i] = prods[i].first; k.b[i] = prods[i].second; }
    cs.push_back(k);
  };
  add("c=1 + 0.75ulp", 1.f, {{ldexpf(1.f, -12), ldexpf(1.5f, -12)}});
  add("c=1 + 0.5ulp (tie)", 1.f, {{ldexpf(1.f, -12), ldexpf(1.f, -12)}});
  add("c=1+ulp + 0.5ulp (tie, odd)", 1.f + u, {{ldexpf(1.f, -12), ldexpf(1.f, -12)}});
  add("c=1 + 0.5078ulp", 1.f, {{ldexpf(1.f, -12), ldexpf(1.015625f, -12)}});
  add("c=1 + 0.99ulp", 1.f, {{ldexpf(1.f, -12), ldexpf(1.984375f, -12)}});
  add("c=-1 + 0.75ulp", -1.f, {{ldexpf(1.f, -12), ldexpf(1.5f, -12)}});
  add("c=-1 - 0.75ulp", -1.f, {{-ldexpf(1.f, -12), ldexpf(1.5f, -12)}});
  add("c=1 - 0.25ulp", 1.f, {{-ldexpf(1.f, -13), ldexpf(1.f, -12)}});
  {  // 16 products of 1/8 ulp each: exact sum 2 ulp
    std::vector<std::pair<float, float>> p(16, {ldexpf(1.f, -13), ldexpf(1.f, -13)});
    add("c=1 + 16 x 0.125ulp (=2ulp)", 1.f, p);
  }
  {  // 16 products of 0.046875 ulp: exact sum 0.75 ulp
    std::vector<std::pair<float, float>> p(16, {ldexpf(1.f, -13), ldexpf(1.5f, -15)});
    add("c=1 + 16 x 0.046875ulp (=0.75ulp)", 1.f, p);
  }
  {  // c = 0, one product 1.0 and 15 of 0.25 ulp: exact 1 + 3.75ulp
    std::vector<std::pair<float, float>> p(16, {ldexpf(1.f, -12), ldexpf(1.f, -13)});
    p[0] = {1.f, 1.f};
    add("c=0: 1 + 15 x 0.25ulp (=3.75ulp)", 0.f, p);
  }
  {  // c = 0: 1 + 15 x 2^-10 ulp (lost entirely unless summed wide): exact 1 + 0.0146ulp
    std::vector<std::pair<float, float>> p(16, {ldexpf(1.f, -16), ldexpf(1.f, -17)});
    p[0] = {1.f, 1.f};
    add("c=0: 1 + 15 x 2^-10ulp", 0.f, p);
  }
  add("c=2^24 + 1.0 (0.5ulp tie)", ldexpf(1.f, 24), {{1.f, 1.f}});
  add("c=2^24 + 1.5", ldexpf(1.f, 24), {{1.f, 1.5f}});
  add("c=2^24 + 1 + 0.5 (two products)", ldexpf(1.f, 24), {{1.f, 1.f}, {1.f, 0.5f}});
  add("c=2^24 + 0.75 + 0.75 (two products =1.5)", ldexpf(1.f, 24), {{1.f, 0.75f}, {1.f, 0.75f}});
  add("c=2^24 + 8 x 0.25 (=2 = 1ulp)", ldexpf(1.f, 24), {{1.f, .25f}, {1.f, .25f}, {1.f, .25f}, {1.f, .25f}, {1.f, .25f}, {1.f, .25f}, {1.f, .25f}, {1.f, .25f}});
  add("c=2^24 + 16 x 2^-5 (=0.5)", ldexpf(1.f, 24), std::vector<std::pair<float, float>>(16, {1.f, ldexpf(1.f, -5)}));
  add("c=2^24 + 16 x 0.09375 (=1.5)", ldexpf(1.f, 24), std::vector<std::pair<float, float>>(16, {1.f, 0.09375f}));
  add("c=2^24+2 + 1 (tie to even up)", ldexpf(1.f, 24) + 2.f, {{1.f, 1.f}});

  const int n = (int)cs.size();
  std::vector<float> hA(n * 32 * 16, 0.f), hB(n * 32 * 16, 0.f), hC(n), hD(n), hD32(n);
  for (int c = 0; c < n; ++c) {
    hC[c] = cs[c].c;
    for (int k = 0; k < 16; ++k) { hA[(c * 32 + 0) * 16 + k] = cs[c].a[k]; hB[(c * 32 + 0) * 16 + k] = cs[c].b[k]; }
  }
  float *dA, *dB, *dC, *dD;
  hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, n * 4); hipMalloc(&dD, n * 4);
  hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dC, hC.data(), n * 4, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dC, dD, n);
  hipMemcpy(hD.data(), dD, n * 4, hipMemcpyDeviceToHost);
  probe32<<<1, 64>>>(dA, dB, dC, dD, n);
  hipMemcpy(hD32.data(), dD, n * 4, hipMemcpyDeviceToHost);
  for (int c = 0; c < n; ++c) {
    double exact = cs[c].c;
    for (int k = 0; k < 16; ++k) exact += (double)cs[c].a[k] * (double)cs[c].b[k];
    const float rn = (float)exact;  // RN-even of the exact sum
    printf("%-44s exact %.10g  RN %08x  bf16-mfma %08x (%+.3f ulp vs RN)  fp32-mfma %08x\n", cs[c].name, exact, bits(rn),
           bits(hD[c]), (double)((long long)bits(hD[c]) - (long long)bits(rn)), bits(hD32[c]));
  }
  return 0;
}
