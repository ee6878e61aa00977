/* ORACLE (test infrastructure) -- internal declarations shared by the oracle's C files. */
#ifndef WH_ORACLE_INTERNAL_H
#define WH_ORACLE_INTERNAL_H
#include "wh_oracle.h"
#include <stdlib.h>

typedef unsigned __int128 u128;

extern const u128 WO_P;
void wo_init(void);
static inline void *wo_alloc(size_t n) { void *p = calloc(1, n ? n : 1); if (!p) abort(); return p; }
static inline void wo_free(void *p) { free(p); }

/* field elements are held in Montgomery form, as the reference does (field.rs:378-388) */
u128 f_add(u128, u128);
u128 f_sub(u128, u128);
u128 f_neg(u128);
u128 f_mul(u128, u128);
u128 f_pow(u128, u128);
u128 f_inv(u128);
u128 f_from_int(u128);
u128 f_to_int(u128);
u128 f_root(unsigned l);
int f_decode(const uint8_t b[16], u128 *out);
void f_encode(u128 m, uint8_t b[16]);

typedef struct { uint64_t st[25]; unsigned pos; } shake_t;
void shake_init(shake_t *);
void shake_absorb(shake_t *, const uint8_t *, size_t);
void shake_finalize(shake_t *);
void shake_squeeze(shake_t *, uint8_t *, size_t);
void xof_init(shake_t *, const uint8_t seed[16], const uint8_t *dst, size_t dst_len);
void xof_into_seed(shake_t *, uint8_t out[16]);
void xof_next_vec(shake_t *, u128 *out, size_t n);
void xof_expand(const uint8_t seed[16], const uint8_t *dst, size_t dst_len, const uint8_t *binder,
                size_t binder_len, u128 *out, size_t n);

int fft(u128 *outp, const u128 *inp, size_t n_in, size_t size);
void fft_inv_finish(u128 *outp, size_t size, u128 size_inv);
u128 poly_eval(const u128 *poly, size_t n, u128 at);

/* flp.c */
int flp_prove(const wo_params *p, const u128 *input, const u128 *prove_rand, const u128 *joint_rand, u128 *proof);
int flp_query(const wo_params *p, const u128 *input, const u128 *proof, const u128 *query_rand,
              const u128 *joint_rand, uint32_t num_shares, u128 *verifier);
int flp_decide(const wo_params *p, const u128 *verifier);
int flp_truncate(const wo_params *p, const u128 *input, u128 *out);
int flp_encode(const wo_params *p, const uint64_t *meas, u128 *input);
uint32_t wo_alg_id(const wo_params *p);

#endif
