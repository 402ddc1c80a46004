// Register butterflies of the library's own FFTs (radix 2 / 4 / 8 / 16, 3 / 9, 5 / 25, 7 / 49 and the composites
// 10 / 12 / 15 / 20 / 24 / 32 / 40), templated on the scalar
// type: used by the Hilbert envelope's Stockham passes (mm_hilbert.hip.inc) and by the any-length STFT (mm_anyfft.hip.inc).
#pragma once
#include "mm_common.h"

template <typename T>
struct hb_c { T x, y; };

template <typename T>
__device__ __forceinline__ hb_c<T> hb_mul(hb_c<T> a, hb_c<T> b) {
  hb_c<T> r;
  r.x = a.x * b.x - a.y * b.y;
  r.y = a.x * b.y + a.y * b.x;
  return r;
}
template <typename T>
__device__ __forceinline__ hb_c<T> hb_add(hb_c<T> a, hb_c<T> b) { hb_c<T> r; r.x = a.x + b.x; r.y = a.y + b.y; return r; }
template <typename T>
__device__ __forceinline__ hb_c<T> hb_sub(hb_c<T> a, hb_c<T> b) { hb_c<T> r; r.x = a.x - b.x; r.y = a.y - b.y; return r; }
template <typename T>
__device__ __forceinline__ hb_c<T> hb_mul_mi(hb_c<T> a) { hb_c<T> r; r.x = a.y; r.y = -a.x; return r; }   // * (-i)

template <typename T>
__device__ __forceinline__ void hb_r2(hb_c<T>& a, hb_c<T>& b) {
  const hb_c<T> t = hb_sub(a, b);
  a = hb_add(a, b);
  b = t;
}
// forward DFT-4 in place, natural order out
template <typename T>
__device__ __forceinline__ void hb_r4(hb_c<T>& v0, hb_c<T>& v1, hb_c<T>& v2, hb_c<T>& v3) {
  const hb_c<T> t0 = hb_add(v0, v2), t1 = hb_sub(v0, v2), t2 = hb_add(v1, v3), t3 = hb_mul_mi(hb_sub(v1, v3));
  v0 = hb_add(t0, t2);
  v2 = hb_sub(t0, t2);
  v1 = hb_add(t1, t3);
  v3 = hb_sub(t1, t3);
}

#include "mm_hilbert_tables.inc"

template <int R>
__device__ __forceinline__ const double (*hb_cs())[2] {
  if constexpr (R == 3) return hb_cs3;
  else if constexpr (R == 5) return hb_cs5;
  else if constexpr (R == 7) return hb_cs7;
  else if constexpr (R == 9) return hb_cs9;
  else if constexpr (R == 25) return hb_cs25;
  else if constexpr (R == 10) return hb_cs10;
  else if constexpr (R == 12) return hb_cs12;
  else if constexpr (R == 15) return hb_cs15;
  else if constexpr (R == 20) return hb_cs20;
  else if constexpr (R == 24) return hb_cs24;
  else if constexpr (R == 32) return hb_cs32;
  else if constexpr (R == 40) return hb_cs40;
  else return hb_cs49;
}

// forward DFT of an odd prime number of points, natural order out: with t_m = v[m] + v[P-m], d_m = v[m] - v[P-m],
//   V[u], V[P-u] = (v[0] + sum_m cos(2 pi u m / P) t_m) -+ i (sum_m sin(2 pi u m / P) d_m)
template <typename T, int P>
__device__ __forceinline__ void hb_dft_prime(hb_c<T> (&v)[P]) {
  constexpr int H = (P - 1) / 2;
  const double (*cs)[2] = hb_cs<P>();
  hb_c<T> t[H], d[H], o[P];
#pragma unroll
  for (int m = 1; m <= H; ++m) { t[m - 1] = hb_add(v[m], v[P - m]); d[m - 1] = hb_sub(v[m], v[P - m]); }
  o[0] = v[0];
#pragma unroll
  for (int m = 0; m < H; ++m) o[0] = hb_add(o[0], t[m]);
#pragma unroll
  for (int u = 1; u <= H; ++u) {
    hb_c<T> a = v[0], b;
    b.x = (T)0; b.y = (T)0;
#pragma unroll
    for (int m = 1; m <= H; ++m) {
      const T c = (T)cs[(u * m) % P][0], sn = (T)cs[(u * m) % P][1];
      a.x += c * t[m - 1].x; a.y += c * t[m - 1].y;
      b.x += sn * d[m - 1].x; b.y += sn * d[m - 1].y;
    }
    o[u].x = a.x + b.y; o[u].y = a.y - b.x;            // a - i b
    o[P - u].x = a.x - b.y; o[P - u].y = a.y + b.x;    // a + i b
  }
#pragma unroll
  for (int u = 0; u < P; ++u) v[u] = o[u];
}

// forward DFT of P * P points (P = 3, 5, 7), natural order out: n = P a + b, u = c + P d --
// DFT-P over a for each b, * W_(P P)^(b c), DFT-P over b for each c
template <typename T, int P>
__device__ __forceinline__ void hb_dft_square(hb_c<T> (&v)[P * P]) {
  const double (*cs)[2] = hb_cs<P * P>();
  hb_c<T> w[P][P];                          // w[b][c]
#pragma unroll
  for (int b = 0; b < P; ++b) {
    hb_c<T> tmp[P];
#pragma unroll
    for (int a = 0; a < P; ++a) tmp[a] = v[P * a + b];
    hb_dft_prime<T, P>(tmp);
#pragma unroll
    for (int c = 0; c < P; ++c) {
      if (b * c) {
        hb_c<T> tw;
        tw.x = (T)cs[b * c][0]; tw.y = (T)(-cs[b * c][1]);
        w[b][c] = hb_mul(tmp[c], tw);
      } else {
        w[b][c] = tmp[c];
      }
    }
  }
#pragma unroll
  for (int c = 0; c < P; ++c) {
    hb_c<T> tmp[P];
#pragma unroll
    for (int b = 0; b < P; ++b) tmp[b] = w[b][c];
    hb_dft_prime<T, P>(tmp);
#pragma unroll
    for (int d = 0; d < P; ++d) v[c + P * d] = tmp[d];
  }
}

// forward DFT of R points in registers; V[u] is left in v[hb_perm<R>(u)]
template <int R>
__device__ __forceinline__ constexpr int hb_perm(int u) {
  return R == 16 ? 4 * (u & 3) + (u >> 2) : (R == 8 ? 2 * (u & 3) + (u >> 2) : u);
}
template <typename T, int R>
__device__ __forceinline__ void hb_dft(hb_c<T> (&v)[R]);

// forward DFT of A * B points, natural order out: n = B a + b, u = c + A d -- DFT-A over a for each b, * W_(AB)^(b c)
// (quarter turns as exact swaps), DFT-B over b for each c
template <typename T, int A, int B>
__device__ __forceinline__ void hb_dft_rect(hb_c<T> (&v)[A * B]) {
  constexpr int N = A * B;
  const double (*cs)[2] = hb_cs<N>();
  hb_c<T> w[B][A];                          // w[b][c]
#pragma unroll
  for (int b = 0; b < B; ++b) {
    hb_c<T> tmp[A];
#pragma unroll
    for (int a = 0; a < A; ++a) tmp[a] = v[B * a + b];
    hb_dft<T, A>(tmp);
#pragma unroll
    for (int c = 0; c < A; ++c) {
      const hb_c<T> x = tmp[hb_perm<A>(c)];
      const int e = b * c;                  // < N
      if (e == 0) {
        w[b][c] = x;
      } else if (4 * e == N) {
        w[b][c] = hb_mul_mi(x);
      } else if (2 * e == N) {
        w[b][c].x = -x.x; w[b][c].y = -x.y;
      } else if (4 * e == 3 * N) {
        w[b][c].x = -x.y; w[b][c].y = x.x;
      } else {
        hb_c<T> tw;
        tw.x = (T)cs[e][0]; tw.y = (T)(-cs[e][1]);
        w[b][c] = hb_mul(x, tw);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < A; ++c) {
    hb_c<T> tmp[B];
#pragma unroll
    for (int b = 0; b < B; ++b) tmp[b] = w[b][c];
    hb_dft<T, B>(tmp);
#pragma unroll
    for (int d = 0; d < B; ++d) v[c + A * d] = tmp[hb_perm<B>(d)];
  }
}

template <typename T, int R>
__device__ __forceinline__ void hb_dft(hb_c<T> (&v)[R]) {
  if constexpr (R == 2) {
    hb_r2(v[0], v[1]);
  } else if constexpr (R == 4) {
    hb_r4(v[0], v[1], v[2], v[3]);
  } else if constexpr (R == 8) {
    // n = 2 a + b (a < 4, b < 2), u = c + 4 d: DFT-4 over a for each b, * W8^(b c), DFT-2 over b
    hb_r4(v[0], v[2], v[4], v[6]);
    hb_r4(v[1], v[3], v[5], v[7]);
    const T H = (T)0.70710678118654752440;
    { const hb_c<T> t = v[3]; v[3].x = (t.x + t.y) * H; v[3].y = (t.y - t.x) * H; }     // W8^1
    v[5] = hb_mul_mi(v[5]);                                                               // W8^2
    { const hb_c<T> t = v[7]; v[7].x = (t.y - t.x) * H; v[7].y = -(t.x + t.y) * H; }    // W8^3
    hb_r2(v[0], v[1]); hb_r2(v[2], v[3]); hb_r2(v[4], v[5]); hb_r2(v[6], v[7]);
    // V[c + 4 d] sits in v[2 c + d]
  } else if constexpr (R == 3 || R == 5 || R == 7) {
    hb_dft_prime<T, R>(v);
  } else if constexpr (R == 9) {
    hb_dft_square<T, 3>(v);
  } else if constexpr (R == 25) {
    hb_dft_square<T, 5>(v);
  } else if constexpr (R == 49) {
    hb_dft_square<T, 7>(v);
  } else if constexpr (R == 10) {
    hb_dft_rect<T, 2, 5>(v);
  } else if constexpr (R == 12) {
    hb_dft_rect<T, 4, 3>(v);
  } else if constexpr (R == 15) {
    hb_dft_rect<T, 3, 5>(v);
  } else if constexpr (R == 20) {
    hb_dft_rect<T, 4, 5>(v);
  } else if constexpr (R == 24) {
    hb_dft_rect<T, 8, 3>(v);
  } else if constexpr (R == 32) {
    hb_dft_rect<T, 16, 2>(v);
  } else if constexpr (R == 40) {
    hb_dft_rect<T, 8, 5>(v);
  } else {
    static_assert(R == 16, "radix");
    // n = 4 a + b, u = c + 4 d: DFT-4 over a, * W16^(b c), DFT-4 over b; V[c + 4 d] sits in v[4 c + d]
#pragma unroll
    for (int b = 0; b < 4; ++b) hb_r4(v[b], v[4 + b], v[8 + b], v[12 + b]);
    const T C1 = (T)0.92387953251128675613, S1 = (T)0.38268343236508977173, H = (T)0.70710678118654752440;
    auto rot = [](hb_c<T> a, T c, T s) { hb_c<T> r; r.x = a.x * c + a.y * s; r.y = a.y * c - a.x * s; return r; };  // * (c - i s)
    v[5] = rot(v[5], C1, S1);        // W16^1
    v[6] = rot(v[6], H, H);          // W16^2
    v[7] = rot(v[7], S1, C1);        // W16^3
    v[9] = rot(v[9], H, H);          // W16^2
    v[10] = hb_mul_mi(v[10]);        // W16^4
    v[11] = rot(v[11], -H, H);       // W16^6
    v[13] = rot(v[13], S1, C1);      // W16^3
    v[14] = rot(v[14], -H, H);       // W16^6
    v[15] = rot(v[15], -C1, -S1);    // W16^9
#pragma unroll
    for (int c = 0; c < 4; ++c) hb_r4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
  }
}

