/* Host check over ALL 2^32 inputs: candidates for the deviate mapping of csrc/philox.hpp (u32_to_unit_float) against the
 * reference's  real( dble(u) / (2**32 - 1) )  (Code/RandomNumbersForMC.f95:275-299).
 *   gcc -O2 -o unit_float_check unit_float_check.c && ./unit_float_check        (eight threads, a few seconds)
 * Prints, per candidate, the number of inputs whose float32 differs. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline float reference(uint32_t u) { return (float)((double)u / 4294967295.0); }
static inline float current(uint32_t u) { return (float)((double)u * (1.0 / 4294967295.0)); }   /* I3RC_UNIT_FLOAT = 0 */
static inline float sticky(uint32_t u) {                                                            /* = 1 */
  double d = (double)u; uint64_t b; memcpy(&b, &d, 8); b |= 1ull; memcpy(&d, &b, 8);
  return (float)d * 2.3283064365386963e-10f;
}
static inline float two_ops(uint32_t u) { return (float)u * 2.3283064365386963e-10f; }              /* = 2 */

typedef struct { uint64_t lo, hi, bad[3]; } job;
static void *run(void *p) {
  job *j = p;
  for (uint64_t v = j->lo; v < j->hi; ++v) {
    const uint32_t u = (uint32_t)v;
    const float r = reference(u);
    uint32_t rb, cb; memcpy(&rb, &r, 4);
    float c = current(u); memcpy(&cb, &c, 4); j->bad[0] += cb != rb;
    c = sticky(u); memcpy(&cb, &c, 4); j->bad[1] += cb != rb;
    c = two_ops(u); memcpy(&cb, &c, 4); j->bad[2] += cb != rb;
  }
  return 0;
}
int main(void) {
  enum { T = 8 };
  pthread_t th[T]; job jobs[T];
  for (int t = 0; t < T; ++t) {
    jobs[t] = (job){(uint64_t)t << 29, (uint64_t)(t + 1) << 29, {0, 0, 0}};
    pthread_create(&th[t], 0, run, &jobs[t]);
  }
  uint64_t bad[3] = {0, 0, 0};
  for (int t = 0; t < T; ++t) { pthread_join(th[t], 0); for (int k = 0; k < 3; ++k) bad[k] += jobs[t].bad[k]; }
  printf("inputs 4294967296\n");
  printf("(float)((double)u * (1.0 / 4294967295.0))        [I3RC_UNIT_FLOAT=0, the kernels' mapping]: %llu differ\n", (unsigned long long)bad[0]);
  printf("(float)(sticky((double)u)) * 2^-32                 [I3RC_UNIT_FLOAT=1]: %llu differ\n", (unsigned long long)bad[1]);
  printf("(float)u * 2^-32                                   [I3RC_UNIT_FLOAT=2]: %llu differ (2^25 = 33554432)\n", (unsigned long long)bad[2]);
  return 0;
}
