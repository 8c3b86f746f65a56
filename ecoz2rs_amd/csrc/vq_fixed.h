// vq_fixed.h -- exact fixed-point accumulation helpers shared by kernels and host driver.
// A value x becomes two signed 32-bit limbs (hi, lo) with x ~= (hi*2^31 + lo) * 2^-(sh+31);
// limb sums are 64-bit integers, so any summation order / partition gives identical results.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace e2vq {

__host__ __device__ __forceinline__ void fix2(double x, int sh, int& hi, int& lo)
{
    const double y = ldexp(x, sh);
    const double h = __builtin_rint(y);
    const double l = __builtin_rint(ldexp(y - h, 31));
    hi = (int)h;
    lo = (int)l;
}

// fix2 with the scaling done by one multiplication (scale = 2^sh, exact while 2^sh is a normal double: |sh| <= 1000 is
// what callers check) and both roundings by the 1.5 * 2^52 trick: y + C lies in [2^52, 2^53), where doubles are the
// integers, so the addition IS round-to-nearest-even to an integer -- the value of rint(y) -- and the low mantissa
// word of the sum is that integer in two's complement (C's own low word is zero).  Valid for |y| < 2^51; here
// |y| < 2^30 and |(y - h) 2^31| <= 2^30.  Identical limbs to fix2 for every input (a product that underflows gives
// (0, 0) either way); six full-rate FP64 operations instead of ldexp / rndne / cvt, which run at a fraction of that rate.
__device__ __forceinline__ void fix2_mul(double x, double scale, int& hi, int& lo)
{
    const double C = 6755399441055744.0;  // 1.5 * 2^52
    const double y = x * scale;
    const double yc = y + C;
    hi = __double2loint(yc);
    const double h = yc - C;
    const double zc = (y - h) * 2147483648.0 + C;
    lo = __double2loint(zc);
}

// (sum_hi*2^31 + sum_lo), rounded to the nearest double (ties to even), times 2^-(sh+31)
__host__ __device__ inline double unfix(long long sum_hi, long long sum_lo, int sh)
{
    __int128 total = (__int128)sum_hi * ((__int128)1 << 31) + (__int128)sum_lo;
    if (total == 0) return 0.0;
    const bool neg = total < 0;
    unsigned __int128 u = neg ? (unsigned __int128)0 - (unsigned __int128)total : (unsigned __int128)total;
    const unsigned long long uh = (unsigned long long)(u >> 64), ul = (unsigned long long)u;
    const int msb = uh ? 127 - __builtin_clzll(uh) : 63 - __builtin_clzll(ul);
    double res;
    if (msb <= 52) {
        res = (double)ul;
    } else {
        const int shift = msb - 52;
        unsigned long long mant = (unsigned long long)(u >> shift);
        const unsigned __int128 rem = u & ((((unsigned __int128)1) << shift) - 1);
        const unsigned __int128 half = ((unsigned __int128)1) << (shift - 1);
        if (rem > half || (rem == half && (mant & 1))) mant++;
        res = ldexp((double)mant, shift);
    }
    res = ldexp(res, -(sh + 31));
    return neg ? -res : res;
}

// exponent bound of (min distortion - 1) for a codebook with max L1 norm l1max
__host__ __device__ __forceinline__ int dist_exponent(double maxabs, double l1max)
{
    const double B = maxabs * l1max + 1.0;
    return ilogb(B) + 2;
}

}  // namespace e2vq
