/* orb_sincos -- the normative cos/sin used to steer the rBRIEF pattern.
 *
 * The reference rotates the sampling pattern with
 *     float a = (float)cos(angle), b = (float)sin(angle);
 * (reference src/ORBextractor.cc:124-125), which resolves to the host libm's
 * cosf/sinf.  libm is neither guaranteed correctly rounded nor available on the
 * GPU, and a 1-ulp difference in a/b can flip a cvRound() of a rotated sample
 * coordinate, i.e. a descriptor bit.  SURVEY.md Appendix A.8 therefore pins ONE
 * libm-free definition that the CPU oracle and the HIP kernels both compile:
 * IEEE-754 double +,-,* only, fixed evaluation order, one final rounding to
 * float.  Both sides MUST be compiled with -ffp-contract=off (no FMA fusion).
 *
 * Accuracy: Cody-Waite reduction by pi/2 (33-bit head, so k*head is exact for
 * the |x| < 2^20 inputs this is used on) followed by the classic degree-13/14
 * minimax kernels on [-pi/4, pi/4]; absolute error < 1e-16 before the final
 * rounding, so the float result is the correctly rounded cos/sin except on
 * astronomically rare near-ties.  tests/test_oracle_math.py sweeps every float
 * angle the extractor can produce and reports the disagreement count against
 * glibc cosf/sinf (measured: 0).
 *
 * This header is shared boundary specification (like the BRIEF table), not part
 * of the oracle: it is the only arithmetic the oracle and the product share.
 */
#ifndef ORB_SINCOS_H
#define ORB_SINCOS_H

#if defined(__HIPCC__)
#define ORB_HD __host__ __device__ inline
#else
#define ORB_HD static inline
#endif

ORB_HD void orb_sincos(float x, float* c_out, float* s_out)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HEAD   = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
    const double PIO2_TAIL   = 6.07710050650619224932e-11; /* pi/2 - PIO2_HEAD      */
    const double S1 = -1.66666666666666324348e-01, S2 =  8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 =  2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 =  1.58969099521155010221e-10;
    const double C1 =  4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 =  2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 =  2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;

    const double xd = (double)x;
    const double t  = xd * TWO_OVER_PI;
    const int    k  = (int)(t + (t >= 0.0 ? 0.5 : -0.5));
    const double kd = (double)k;
    double r = xd - kd * PIO2_HEAD;
    r = r - kd * PIO2_TAIL;

    const double z = r * r;
    double ps = S5 + z * S6;
    ps = S4 + z * ps;
    ps = S3 + z * ps;
    ps = S2 + z * ps;
    ps = S1 + z * ps;
    const double sn = r + (r * z) * ps;

    double pc = C5 + z * C6;
    pc = C4 + z * pc;
    pc = C3 + z * pc;
    pc = C2 + z * pc;
    pc = C1 + z * pc;
    const double cs = (1.0 - 0.5 * z) + (z * z) * pc;

    double c, s;
    switch (k & 3) {
    case 0:  c =  cs; s =  sn; break;
    case 1:  c = -sn; s =  cs; break;
    case 2:  c = -cs; s = -sn; break;
    default: c =  sn; s = -cs; break;
    }
    *c_out = (float)c;
    *s_out = (float)s;
}

#endif /* ORB_SINCOS_H */
