/*
 * ctrhip.h -- C ABI of libctrhip.so: hand-written gfx950 (MI355X) kernels for the
 * forward/backward of a CTR model zoo.
 *
 * The reference (WardellZc/DeepLearningRecommendationSystem) has no FFI layer:
 * its hot path is `model/<m>.py: nn.Module.forward` + autograd, called from
 * `trainer/trainer.py:23-40`.  These entry points are what a binding for that
 * path binds instead of ATen ops; each declaration names the reference lines
 * it replaces.  INTEGRATION.md shows the ctypes stub a maintainer adds.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the comment says "host";
 *  - tensors are fp32, row-major, with an explicit leading dimension in
 *    elements (ld*); indices are int64 as `nn.Embedding` takes them;
 *  - `stream` is a hipStream_t; every call only enqueues work on it, never
 *    allocates, never synchronises, keeps no state between calls and is
 *    re-entrant per stream (one caller-owned exception: the ticket word of
 *    ctr_bce_fwd, zero before and after every call);
 *  - return value: 0 = enqueued, <0 = CTR_E* (nothing enqueued).  Never
 *    throws, never exits.  ctr_strerror() names a code;
 *  - out-of-range indices never fault: the row is treated as row 0 and, when
 *    `err_flag` (device int32, may be NULL) is given, it is set to 1 so the
 *    host can raise the IndexError `nn.Embedding` would have raised.
 */
#ifndef CTRHIP_H
#define CTRHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTR_OK 0
#define CTR_EINVAL (-1)   /* bad argument (null pointer, negative size, bad kind) */
#define CTR_ELIMIT (-2)   /* shape outside what the kernels are built for */
#define CTR_ELAUNCH (-3)  /* hipLaunchKernel reported an error */
#define CTR_EALIGN (-4)   /* pointer / leading dimension not aligned as required */

int ctr_version(void);                 /* ABI version, bumped on any signature change */
const char* ctr_strerror(int code);    /* host string, static storage */
const char* ctr_target_arch(void);     /* "gfx950" */

/* ------------------------------------------------------------------------
 * Embedding stage: fused row gather + weighted bag pooling + concat.
 * Replaces the nn.Embedding calls, the "multi-hot matmul" poolings and the
 * torch.cat that follows them:
 *   model/pnn.py:113-121, model/deepfm.py:45-54, model/deepcrossing.py:63-71,
 *   model/ffm.py:48-59, model/neuralcf.py:36-46, model/mf.py:24-25.
 * One launch writes, for every sample b and field f, `width` floats at
 * out[b*ldo + out_col ...].
 * ---------------------------------------------------------------------- */
enum {
  CTR_FIELD_ID_I64 = 0,  /* out = table[idx[b*idx_stride]]                        (K1/K3) */
  CTR_FIELD_ID_F32 = 1,  /* out = table[(int64)x[b*ldx+src_col]]  ("x[:,c].long()") (K1) */
  CTR_FIELD_BAG = 2,     /* out = sum_j x[b*ldx+src_col+j] * table[j], j<bag_size  (K2) */
  CTR_FIELD_DENSE = 3,   /* out = x[b*ldx+src_col .. +width)   (deepcrossing.py:65)      */
  CTR_FIELD_PROD_I64 = 4 /* out = table[idx[b]] * table2[idx2[b]] (neuralcf.py:36-39)    */
};
#define CTR_MAX_FIELDS 32

typedef struct ctr_field {
  int32_t kind;
  int32_t width;          /* floats written per sample (embedding dim E) */
  int32_t out_col;        /* first output column */
  int32_t src_col;        /* ID_F32 / BAG / DENSE: first column of x */
  int32_t bag_size;       /* BAG: K (table has K rows) */
  int32_t reserved;
  int64_t vocab;          /* rows of `table` (bounds check) */
  int64_t idx_stride;     /* ID_I64 / PROD: elements between consecutive samples */
  const int64_t* idx;     /* ID_I64 / PROD */
  const float* table;     /* (vocab, width) */
  float* grad;            /* backward: dense (vocab, width) accumulator, NULL = skip */
  const int64_t* idx2;    /* PROD only */
  const float* table2;    /* PROD only, (vocab2, width) */
  float* grad2;           /* PROD only */
  int64_t vocab2;         /* PROD only */
} ctr_field_t;

/* fields: host array of nfields (<= CTR_MAX_FIELDS) descriptors, copied by value */
int ctr_embed_fwd(const ctr_field_t* fields, int nfields, const float* x, int64_t ldx,
                  int64_t batch, float* out, int64_t ldo, int32_t* err_flag, void* stream);

/* backward of ctr_embed_fwd (autograd of the lines above; embedding_dense_backward
 * and the `x^T g` matmul backward, trainer/trainer.py:38): accumulates
 * (+=) into each field's dense `grad`; the caller zero-fills it first when it
 * wants a fresh gradient.  Id rows: fp32 atomics (order of accumulation not
 * fixed).  Bag tables (every sample hits the same few rows): reduced per
 * workgroup in LDS, then through `workspace` (device scratch, nullable; see
 * ctr_linear_bwd) or, without it, by atomics. */
int ctr_embed_bwd(const ctr_field_t* fields, int nfields, const float* x, int64_t ldx,
                  int64_t batch, const float* gout, int64_t ldo,
                  float* workspace, int64_t workspace_floats, void* stream);
/* ------------------------------------------------------------------------
 * Matrix factorisation, fused (model/mf.py:23-26):
 *   prob[b] = sigmoid(sum_e U[u[b],e] * V[i[b],e])
 * bwd: dlogit = gprob*prob*(1-prob); dU[u[b]] += dlogit*V[i[b]]; dV likewise.
 * ---------------------------------------------------------------------- */
int ctr_mf_fwd(const float* user_table, int64_t num_users, const float* item_table, int64_t num_items,
               int dim, const int64_t* user_idx, const int64_t* item_idx, int64_t batch,
               float* prob, int32_t* err_flag, void* stream);
int ctr_mf_bwd(const float* user_table, int64_t num_users, const float* item_table, int64_t num_items,
               int dim, const int64_t* user_idx, const int64_t* item_idx, int64_t batch,
               const float* prob, const float* gprob, float* guser, float* gitem, void* stream);

/* ------------------------------------------------------------------------
 * Dense layers (every nn.Linear of the zoo, e.g. model/neuralcf.py:48-51,57,
 * model/pnn.py:18-23,75-76, model/deepfm.py:57-60, model/din.py:14-29).
 *   Y[m, n] = act( sum_k X[m,k] * W[n,k] + bias[n] (+ R[m,n]) )
 * W is nn.Linear's (out_features, in_features) layout.  fp32 in, fp32 MFMA
 * accumulate (v_mfma_f32_32x32x2_f32), fp32 out.
 * ---------------------------------------------------------------------- */
enum { CTR_ACT_NONE = 0, CTR_ACT_RELU = 1, CTR_ACT_SIGMOID = 2 };

int ctr_linear_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias /*nullable*/,
                   const float* residual /*nullable*/, int64_t ldr,
                   float* y, int64_t ldy, int64_t m, int n, int k, int act, void* stream);

/* gz = gy * act'(y) (uses the saved OUTPUT y: relu mask y>0, sigmoid y(1-y));
 * gx[m,k] = sum_n gz[m,n] W[n,k]      (skipped when gx == NULL;  += when accumulate_gx)
 * gw[n,k] += sum_m gz[m,n] X[m,k]     (skipped when gw == NULL)
 * gb[n]  += sum_m gz[m,n]             (skipped when gb == NULL; needs gw)
 * Autograd of the nn.Linear + activation lines above (trainer/trainer.py:38).
 * gw/gb are sums over row chunks computed by different workgroups.  With a
 * `workspace` (device scratch, `workspace_floats` floats, may be NULL; 16M floats
 * give every layer of the zoo its full parallelism) each chunk stores its partial there and a second
 * pass adds them in a fixed order (reproducible); without it the chunks accumulate
 * with fp32 atomics.  Either way the result is ADDED to gw/gb: zero-fill for a
 * fresh gradient. */
int ctr_linear_bwd(const float* x, int64_t ldx, const float* w, int64_t ldw,
                   const float* y, int64_t ldy, const float* gy, int64_t ldgy,
                   float* gx, int64_t ldgx, int accumulate_gx,
                   float* gw, int64_t ldgw, float* gb,
                   int64_t m, int n, int k, int act,
                   float* workspace, int64_t workspace_floats, void* stream);

/* ------------------------------------------------------------------------
 * Feature interactions on the stacked embedding matrix emb (batch, >= nvec*dim)
 * written by ctr_embed_fwd: vector f of sample b is emb[b*lde + f*dim ...].
 * ---------------------------------------------------------------------- */

/* PNN inner products (model/pnn.py:59-66): out[b, idx(i,j)] = <v_i, v_j>, i<j in
 * lexicographic order, nvec*(nvec-1)/2 columns. */
int ctr_allpairs_fwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim,
                     float* out, int64_t ldo, void* stream);
/* gemb[b,i,:] (= or +=) sum_{j!=i} gp[b, idx(i,j)] * v_j */
int ctr_allpairs_bwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim,
                     const float* gp, int64_t ldgp, float* gemb, int64_t ldg, int accumulate, void* stream);

/* NFM bi-interaction pooling (model/nfm.py:56-61): out[b, e] = sum_{i<j} v_i[e] * v_j[e], pairs in
 * the reference's order; emb as for ctr_allpairs_fwd.  bwd: gemb[b,i,e] (= or +=) g[b,e] * sum_{j!=i} v_j[e]. */
int ctr_biinteract_fwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim,
                       float* out, int64_t ldo, void* stream);
int ctr_biinteract_bwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim,
                       const float* gout, int64_t ldgo, float* gemb, int64_t ldg, int accumulate, void* stream);

/* AFM pair products (model/afm.py:56-60): out[(b*np + idx(i,j))*ldo + e] = v_i[e] * v_j[e], i<j
 * lexicographic, np = nvec*(nvec-1)/2 rows per sample, nvec <= 16.
 * bwd: gpair = gp (+ attn[b*np + p] * gpool[b] when both are given: the attention-weighted sum
 * of afm.py:65); gemb[b,i,e] (= or +=) sum_{j!=i} gpair[b, idx(i,j), e] * v_j[e]. */
int ctr_pairprod_fwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim,
                     float* out, int64_t ldo, void* stream);
int ctr_pairprod_bwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim,
                     const float* gp, int64_t ldgp, const float* attn /*nullable*/,
                     const float* gpool /*nullable*/, int64_t ldgo,
                     float* gemb, int64_t ldg, int accumulate, void* stream);

/* N single-id fields, gather fused with the FM interaction (model/deepfm.py:45-46 applied to each of
 * F id columns, :63 first-order id terms, :71-77 second order; BASELINE configs[2] "26 fields x 1e6
 * vocab").  idx is the (batch, >= nfields) int64 id matrix (row stride ldidx); tables[f] is the
 * (vocabs[f], dim) table of field f, first[f] its (vocabs[f], 1) first-order weight (first or any
 * first[f] may be NULL); host arrays of device pointers.  dim in {8, 16, 32, 64}.
 *   emb[b, f*dim + e] = tables[f][idx[b,f], e]                       (bit-exact row copies)
 *   fm[b*ldfm] = (sum_f first[f][idx[b,f]] + bias[0]) + 0.5 * sum_e[(sum_f v_fe)^2 - sum_f v_fe^2]
 * A bad id reads row 0 and raises *err_flag (nullable). */
int ctr_fields_fm_fwd(const int64_t* idx, int64_t ldidx, int64_t batch, int nfields, int dim,
                      const float* const* tables, const int64_t* vocabs, const float* const* first /*nullable*/,
                      const float* bias /*nullable*/, float* emb, int64_t lde, float* fm, int64_t ldfm,
                      int32_t* err_flag, void* stream);
/* backward of the above for g = gfm[b*ldgfm] and the gradient gdeep (batch, nfields*dim) that the
 * consumers of emb returned (either may be NULL): dense scatter-add with fp32 atomics
 *   gtables[f][idx[b,f], e] += gdeep[b, f*dim+e] + g * (S_e - v_fe)   (v read back from emb, S_e = sum_f v_fe)
 *   gfirst[f][idx[b,f]] += g,   gbias[0] += sum_b g   (fixed-order partials through the workspace)
 * gtables[f] / gfirst / gfirst[f] / gbias may be NULL (frozen). */
int ctr_fields_fm_bwd(const int64_t* idx, int64_t ldidx, int64_t batch, int nfields, int dim, const int64_t* vocabs,
                      const float* emb, int64_t lde, const float* gdeep /*nullable*/, int64_t ldg,
                      const float* gfm /*nullable*/, int64_t ldgfm, float* const* gtables,
                      float* const* gfirst /*nullable*/, float* gbias /*nullable*/, float* workspace,
                      int64_t workspace_floats, void* stream);

/* the same two operations (same arguments, same results) for many vectors -- PNN over F id fields, 325 products at
 * F = 26 (model/pnn.py:59-66 applied to F fields).  The pinned shape F = 26, dim = 16 with 16-byte aligned, 4-float
 * padded prod / gp rows runs with all 26 vectors of a sample in the registers of a lane quad (forward 45 us,
 * backward 102 us at batch 65536 against 132 / 275 us of ctr_allpairs_*).  Other shapes: forward through a
 * wave-private LDS strip (dim in {8, 16, 32, 64}, F*dim floats x 256/dim samples <= 56 KB); backward returns
 * CTR_ELIMIT with nothing enqueued and the caller uses ctr_allpairs_bwd. */
int ctr_fields_pairs_fwd(const float* emb, int64_t lde, int64_t batch, int nfields, int dim,
                         float* prod, int64_t ldp, void* stream);
int ctr_fields_pairs_bwd(const float* emb, int64_t lde, int64_t batch, int nfields, int dim,
                         const float* gp, int64_t ldgp, float* gemb, int64_t ldg, int accumulate, void* stream);

/* DeepFM wide part + FM second order (model/deepfm.py:63,71-77):
 *   out[b*ldo] = user1[u] + item1[i] + (x[b, dense_col0..+ndense) . wide_w + wide_b)
 *               + 0.5 * sum_e[(sum_f v_fe)^2 - sum_f v_fe^2]
 * with u = (int64)x[b,user_col], i = (int64)x[b,item_col]; user1/item1 are the
 * (vocab,1) tables `self.user` / `self.item`, wide_w the (1,ndense) weight. */
int ctr_fm_wide_fwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim,
                    const float* x, int64_t ldx, int user_col, int item_col, int dense_col0, int ndense,
                    const float* user1, int64_t num_users, const float* item1, int64_t num_items,
                    const float* wide_w, const float* wide_b, float* out, int64_t ldo,
                    int32_t* err_flag, void* stream);
/* g = gout[b*ldgo]: guser1[u] += g, gitem1[i] += g, gwide_w += sum_b g x, gwide_b += sum_b g,
 * gemb[b,f,e] (= or +=) g * (sum_f' v_f'e - v_fe).  Any grad pointer may be NULL. */
int ctr_fm_wide_bwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim,
                    const float* x, int64_t ldx, int user_col, int item_col, int dense_col0, int ndense,
                    const float* user1, int64_t num_users, const float* item1, int64_t num_items,
                    const float* wide_w, const float* wide_b, const float* gout, int64_t ldgo,
                    float* guser1, float* gitem1, float* gwide_w, float* gwide_b,
                    float* gemb, int64_t ldg, int accumulate,
                    float* workspace, int64_t workspace_floats, void* stream);

/* FFM head (model/ffm.py:62-86): cross = sum_p <v_a(p), v_b(p)> over the host pair
 * list `pairs` (2*npairs ints, npairs <= 64), summed left to right;
 *   prob[b*ldo] = sigmoid(user1[u] + item1[i] + sum_c (x[b,c] + cross) * lin_w[c] + lin_b)
 * -- the reference adds the cross scalar to every dense input before its linear
 * layer (ffm.py:84-86) and so does this. */
int ctr_ffm_head_fwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim,
                     const int32_t* pairs, int npairs,
                     const float* x, int64_t ldx, int user_col, int item_col, int dense_col0, int ndense,
                     const float* user1, int64_t num_users, const float* item1, int64_t num_items,
                     const float* lin_w, const float* lin_b, float* prob, int64_t ldo,
                     int32_t* err_flag, void* stream);
int ctr_ffm_head_bwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim,
                     const int32_t* pairs, int npairs,
                     const float* x, int64_t ldx, int user_col, int item_col, int dense_col0, int ndense,
                     const float* user1, int64_t num_users, const float* item1, int64_t num_items,
                     const float* lin_w, const float* lin_b,
                     const float* prob, int64_t ldp, const float* gprob, int64_t ldgp,
                     float* guser1, float* gitem1, float* glin_w, float* glin_b,
                     float* gemb, int64_t ldg,
                     float* workspace, int64_t workspace_floats, void* stream);

/* out (= or +=) gy * act'(y): the residual branch of model/deepcrossing.py:26 */
int ctr_act_bwd(const float* y, int64_t ldy, const float* gy, int64_t ldgy, float* out, int64_t ldo,
                int64_t m, int n, int act, int accumulate, void* stream);

/* ------------------------------------------------------------------------
 * Cross layer of Deep & Cross (model/deepcross.py:7-18):
 *   x_{l+1} = x0 * (W_l x_l) + b_l + x_l,  W_l a bias-free (d, d) nn.Linear.
 * u = x_l W_l^T is a ctr_linear_fwd call; these are the combine around it.
 *   fwd: y[m,d] = x0 * u + bias + xl
 *   bwd: gu = gy * x0 (operand of the layer's dX / dW, i.e. ctr_linear_bwd with gy := gu),
 *        gx0 += gy * u, gbias += column sums of gy (through `workspace`, fixed order).
 *        The gradient w.r.t. x_l is gy + gu W_l: accumulate the dX of ctr_linear_bwd
 *        (accumulate_gx) into the gy buffer.   d <= 1024.
 * ---------------------------------------------------------------------- */
int ctr_cross_fwd(const float* x0, int64_t ldx0, const float* u, int64_t ldu, const float* xl, int64_t ldxl,
                  const float* bias, float* y, int64_t ldy, int64_t m, int d, void* stream);
int ctr_cross_bwd(const float* x0, int64_t ldx0, const float* u, int64_t ldu, const float* gy, int64_t ldgy,
                  float* gu, int64_t ldgu, float* gx0, int64_t ldgx0, float* gbias, int64_t m, int d,
                  float* workspace, int64_t workspace_floats, void* stream);

/* ------------------------------------------------------------------------
 * DIN / DIEN attention over the behaviour sequence (model/din.py:33-53,
 * model/dien.py:23-39).  hist is (batch, len) int64 row-major, target (batch,).
 * ---------------------------------------------------------------------- */

/* K3 sequence gather fused with the attention operand (din.py:35-42):
 *   CTR_DIN_TRIPLE: c[(b*len+l)*ldc + 0:E] = h, [E:2E] = h - t, [2E:3E] = t   (the reference's cat)
 *   CTR_DIN_PAIR:   c[(b*len+l)*ldc + 0:E] = h, [E:2E] = t.  Since W.[h, h-t, t] =
 *                   (Wa+Wb).h + (Wc-Wb).t, a caller that folds the first attention layer's weight
 *                   columns this way gets the same layer output from a 2E-wide operand (2/3 of
 *                   the bytes and flops of the largest GEMM of the step).
 *   h = table[hist[b,l]], t = table[target[b]];  tvec[b*ldt + 0:E] = t  (tvec may be NULL). */
enum { CTR_DIN_TRIPLE = 0, CTR_DIN_PAIR = 1, CTR_DIN_H = 2 /* forward only: c[(b*len+l)*ldc + 0:E] = h, the E-wide operand */ };
int ctr_din_concat_fwd(const float* table, int64_t vocab, int dim, const int64_t* hist, const int64_t* target,
                       int64_t batch, int len, float* c, int64_t ldc, float* tvec, int64_t ldt, int layout,
                       int32_t* err_flag, void* stream);
/* attn[b,:] = softmax_len(score[b,:]) with no padding mask (din.py:44); h_l is read
 * from hsrc[(b*len+l)*ldh ...].  summed != 0: out[b*ldo ...] = sum_l attn_l h_l
 * (din.py:47); summed == 0: out[(b*len+l)*ldo ...] = attn_l h_l (dien.py:37). */
int ctr_din_pool_fwd(const float* score, const float* hsrc, int64_t ldh, int64_t batch, int len, int dim,
                     float* attn, float* out, int64_t ldo, int summed, void* stream);
/* gradient of the scores through pooling + softmax; gout as `out` above */
int ctr_din_pool_bwd(const float* attn, const float* hsrc, int64_t ldh, int64_t batch, int len, int dim,
                     const float* gout, int64_t ldgo, int summed, float* gscore, void* stream);
/* every gradient path into the dense item-table gradient (+=, fp32 atomics):
 *   CTR_DIN_TRIPLE: row hist[b,l] += gc[.,0:E] + gc[.,E:2E] + attn[b,l] * gout_l
 *                   row target[b] += sum_l (gc[.,2E:3E] - gc[.,E:2E]) + gt_extra[b]
 *   CTR_DIN_PAIR:   row hist[b,l] += gc[.,0:E] + attn[b,l] * gout_l
 *                   row target[b] += sum_l gc[.,E:2E] + gt_extra[b]          (gt_extra may be NULL) */
int ctr_din_concat_bwd(const int64_t* hist, const int64_t* target, int64_t vocab, int64_t batch, int len, int dim,
                       const float* gc, int64_t ldc, const float* attn, const float* gout, int64_t ldgo,
                       int summed, const float* gt_extra, int64_t ldgt, int layout, float* gtable, void* stream);

/* ------------------------------------------------------------------------
 * DIEN interest evolution: nn.GRU(E, E, batch_first=True), one layer, h0 = 0
 * (model/dien.py:47,61), dim <= 64.  gi = X W_ih^T + b_ih for all steps is a
 * ctr_linear_fwd call; these run the recurrence.
 *   hbuf: (batch, len+1, dim), hbuf[b,0,:] = 0, hbuf[b,t+1,:] = h_t; `last`
 *   (nullable) receives hidden[-1] = h_{len-1} at last[b*ldl ...].
 * Backward writes dgi (batch*len, 3*dim) and dgh (batch, len+1, 3*dim; row 0 zero,
 * row t+1 = gradient of W_hh h_{t-1} + b_hh), from which
 *   dW_ih = dgi^T X, db_ih = sum dgi, dX = dgi W_ih,
 *   dW_hh = dgh[1:]^T hbuf[:-1], db_hh = sum dgh       are ctr_linear_bwd calls.
 * ---------------------------------------------------------------------- */
int ctr_gru_fwd(const float* gi, int64_t ldgi, const float* w_hh, const float* b_hh, int64_t batch, int len,
                int dim, float* hbuf, float* last, int64_t ldl, void* stream);
int ctr_gru_bwd(const float* gi, int64_t ldgi, const float* w_hh, const float* b_hh, const float* hbuf,
                int64_t batch, int len, int dim, const float* glast, int64_t ldgl, float* dgi, float* dgh,
                void* stream);
/* The same GRU with the input projection inside: x (batch*len, dim) rows instead of gi, W_ih (3*dim, dim), b_ih.
 * Forward writes hbuf / last as ctr_gru_fwd.  Backward recomputes the projection and leaves, instead of dgi / dgh,
 *   gx (batch*len, dim) = dgi W_ih,   gw_ih += dgi^T X,  gb_ih += sum dgi,  gw_hh += dgh^T H_prev,  gb_hh += sum dgh
 * (weight sums per workgroup in registers, fixed-order partials through the workspace: >= 256 * 1632 floats).
 * dim == 16 only (CTR_ELIMIT otherwise with nothing enqueued -- use ctr_linear_fwd + ctr_gru_fwd / ctr_gru_bwd +
 * ctr_linear_bwd as above).  With 16-byte aligned x / hbuf / gx rows (ldx, ldgx multiples of 4) the recurrence runs on
 * the matrix cores, sixteen samples per wave, for any batch; otherwise on DPP rows of four samples, batch % 4 == 0. */
int ctr_gru_fused_fwd(const float* x, int64_t ldx, const float* w_ih, const float* b_ih, const float* w_hh,
                      const float* b_hh, int64_t batch, int len, int dim, float* hbuf, float* last /*nullable*/,
                      int64_t ldl, void* stream);
int ctr_gru_fused_bwd(const float* x, int64_t ldx, const float* w_ih, const float* b_ih, const float* w_hh,
                      const float* b_hh, const float* hbuf, int64_t batch, int len, int dim, const float* glast,
                      int64_t ldgl, float* gx, int64_t ldgx, float* gw_ih, float* gb_ih, float* gw_hh, float* gb_hh,
                      float* workspace, int64_t workspace_floats, void* stream);

/* ------------------------------------------------------------------------
 * A whole stack of narrow nn.Linear(+activation) layers in one launch (forward) and
 * one launch (backward): e.g. NeuralCF's tower, model/neuralcf.py:48-51.  A wave
 * carries 32 rows through every layer with the activations in LDS and all weights
 * of the stack resident in LDS.  Limits: <= 8 layers, every k a multiple of 8 and
 * <= 128, every n <= 128, k[i] == n[i-1], weights 16-byte aligned, intermediate
 * outputs 16-byte aligned with ldy % 4 == 0.  Anything else returns CTR_ELIMIT /
 * CTR_EALIGN without enqueueing; use ctr_linear_* layer by layer then.
 * ---------------------------------------------------------------------- */
typedef struct ctr_mlp_layer {
  const float* w;   /* (n, k) row-major */
  const float* b;   /* (n) or NULL */
  float* y;         /* forward: output (m, ldy), kept for backward; backward: read */
  int64_t ldy;
  float* gw;        /* backward: (n, k), accumulated into (+=) */
  float* gb;        /* backward: (n),   accumulated into (+=) */
  int32_t n, k, act;
  int32_t reserved;
} ctr_mlp_layer_t;

/* y_0 = act_0(x W_0^T + b_0), y_i = act_i(y_{i-1} W_i^T + b_i); layers: host array */
int ctr_mlp_fwd(const float* x, int64_t ldx, int64_t m, const ctr_mlp_layer_t* layers, int nlayers,
                void* stream);
/* The same stack with a single-unit head formed in its epilogue, while the last activations are still
 * on chip:  out[row*ldout] = act( x_extra[row, 0:p] . w[0:p] + y_last[row, :] . w[p:] + c[0] ).
 * NeuralCF's folded linear2 (see ctr_fold_head_fwd): x_extra = the GMF product, w = wfold, c = cfold.
 * The stack's own outputs are written as by ctr_mlp_fwd.  Limits: p <= 64 and a multiple of 8, x_extra
 * rows 16-byte aligned (else CTR_ELIMIT / CTR_EALIGN, nothing enqueued). */
typedef struct ctr_mlp_head {
  const float* x;   /* (m, ldx): p extra input columns, or NULL when p == 0 */
  int64_t ldx;
  const float* w;   /* p + n_last weights */
  const float* c;   /* one bias (device) */
  float* out;       /* (m) at stride ldout */
  int64_t ldout;
  int32_t p, act;
} ctr_mlp_head_t;
int ctr_mlp_head_fwd(const float* x, int64_t ldx, int64_t m, const ctr_mlp_layer_t* layers, int nlayers,
                     const ctr_mlp_head_t* head, void* stream);
/* The embedding stage and the stack behind it in one launch: the effect of
 *   ctr_embed_fwd(fields, nfields, NULL, 0, batch, out, ldo, err_flag)  followed by
 *   ctr_mlp_head_fwd(out, ldo, batch, layers, nlayers, head)            with head->x inside `out`,
 * for the pattern the library has a kernel for -- NeuralCF at BASELINE configs[1] (model/neuralcf.py:37-56:
 * fields = [ID_I64 64 -> column 0, ID_I64 64 -> column 64, PROD_I64 64 -> column 128] of one matrix, the pinned
 * 128-64-32-16-8 ReLU tower on columns 0..127, head->x = out + 128, p = 64).  The kernel reads the ids and table rows
 * itself.  The product columns (head->x) are always written; the tower's input columns 0..127 only with
 * write_x != 0 -- with write_x == 0 they stay untouched and the backward must be ctr_embed_mlp_head_bwd, which
 * gathers them again.  Any other shape: CTR_ELIMIT, nothing enqueued -- issue the two calls. */
/* optional: ctr_fold_head_fwd's map done by the same launch (NULL: head->w / head->c are read as they are).  The
 * kernel forms wfold = [u_full[:p] | W^T u_full[p:]] and cfold = b . u_full[p:] + b2 itself and also writes them to
 * `wfold` / `cfold`, which must be the buffers head->w / head->c point at (the backward reads them there). */
typedef struct ctr_head_fold {
  const float* u_full;  /* (p + n) */
  const float* w;       /* (n, k) at row stride ldw */
  int64_t ldw;
  const float* b;       /* (n) or NULL */
  const float* b2;      /* (1) or NULL */
  float* wfold;         /* (p + k), == head->w */
  float* cfold;         /* (1),     == head->c */
  int32_t p, n, k, reserved;
} ctr_head_fold_t;
int ctr_embed_mlp_head_fwd(const ctr_field_t* fields, int nfields, int64_t batch, float* out, int64_t ldo,
                           int32_t* err_flag /*nullable*/, int write_x, const ctr_mlp_layer_t* layers, int nlayers,
                           const ctr_mlp_head_t* head, const ctr_head_fold_t* fold /*nullable*/, void* stream);
/* Backward of ctr_mlp_head_fwd in ONE launch with the stack's backward: per row gz = gprob * act'(prob);
 * the stack's gY is gz * w[p:] (never stored), gx_extra[row, 0:p] = gz * w[0:p] is written, and
 * gw[0:p+n_last] += sum_rows gz * [x_extra | y_last],  gc[0] += sum_rows gz.  Layers' gw / gb and gx as in
 * ctr_mlp_bwd.  Available for the pinned NeuralCF tower (128-64-32-16-8, relu) with p == 64 only; any other
 * stack returns CTR_ELIMIT without enqueueing (then: ctr_linear_bwd on the head + ctr_mlp_bwd).
 * workspace: (workgroups <= 256) * (sum_i (n_i*k_i + n_i) + p + n_last + 1) floats. */
typedef struct ctr_mlp_head_grad {
  const float* prob;  int64_t ldprob;   /* head output (m), as written by ctr_mlp_head_fwd */
  const float* gprob; int64_t ldgprob;  /* its gradient (m) */
  const float* x;     int64_t ldx;      /* the p extra input columns */
  const float* w;                       /* p + n_last head weights */
  float* gx;          int64_t ldgx;     /* gradient of the extra columns, written (=) */
  float* gw;                            /* p + n_last, accumulated */
  float* gc;                            /* 1, accumulated */
  int32_t p, act;
} ctr_mlp_head_grad_t;
int ctr_mlp_head_bwd(const float* x, int64_t ldx, int64_t m, const ctr_mlp_layer_t* layers, int nlayers,
                     const ctr_mlp_head_grad_t* head, float* gx, int64_t ldgx, float* workspace,
                     int64_t workspace_floats, void* stream);
/* ctr_mlp_head_bwd for a forward done by ctr_embed_mlp_head_fwd(..., write_x = 0, ...): the stack's input
 * [table[idx] | table[idx]] of the first two fields is gathered again by the samples' ids (out-of-range ids read
 * row 0, as in the forward) instead of being read from memory; hg->x is the product columns the forward wrote.
 * Same outputs and workspace contract as ctr_mlp_head_bwd.  CTR_ELIMIT: not the pattern, nothing enqueued. */
/* optional (NULL: hg->gw / hg->gc only): ctr_fold_head_bwd done by the reduction launch of the same call, from the
 * totals hg->gw (p + k) / hg->gc hold afterwards -- same arguments and (+=) semantics as ctr_fold_head_bwd. */
typedef struct ctr_head_fold_grad {
  const float* u_full;  /* (p + n) */
  const float* w;       /* (n, k) at row stride ldw */
  int64_t ldw;
  const float* b;       /* (n) or NULL */
  float* gu_full;       /* (p + n) or NULL */
  float* gw;            /* (n, k) at row stride ldgw, or NULL */
  int64_t ldgw;
  float* gb;            /* (n) or NULL */
  float* gb2;           /* (1) or NULL */
  int32_t p, n, k, reserved;
} ctr_head_fold_grad_t;
/* zero_buf (nullable; 16-byte aligned, zero_floats a multiple of 4): cleared by the first launch of the call, before
 * anything accumulates -- the buffer the (+=) outputs of this step live in (torch's optimizer.zero_grad / the fill
 * launch in front of the backward).  On CTR_ELIMIT nothing was enqueued and nothing was cleared. */
int ctr_embed_mlp_head_bwd(const ctr_field_t* fields, int nfields, int64_t batch, const ctr_mlp_layer_t* layers,
                           int nlayers, const ctr_mlp_head_grad_t* hg, const ctr_head_fold_grad_t* fold /*nullable*/,
                           float* gx, int64_t ldgx, float* workspace, int64_t workspace_floats,
                           float* zero_buf /*nullable*/, int64_t zero_floats, void* stream);
/* ------------------------------------------------------------------------
 * A first Linear on a concatenation of two gathered rows, with the layer applied to the TABLE ROWS (any width; the
 * projected tables A = T_a W_a^T, B = T_b W_b^T + bias are two ctr_linear_fwd calls over the tables):
 *   out[b, :] = act(A[idx_a[b], :] + B[idx_b[b], :])                (model/neuralcf.py:43-49 for any tower)
 * An id outside its table reads row 0 and raises *err_flag.  Backward: ctr_act_mask_bwd turns the gradient of `out`
 * into the gradient of the pre-activation in place (g *= act'(out)); ctr_embed_bwd with two id fields on the same
 * gradient columns forms the row sums S_a, S_b; ctr_linear_bwd over the table rows turns them into the tables' and the
 * layer's gradients (dT = S W, dW = S^T T). */
int ctr_rows_sum_act_fwd(const float* table_a, const int64_t* idx_a, int64_t stride_a, int64_t vocab_a,
                         const float* table_b, const int64_t* idx_b, int64_t stride_b, int64_t vocab_b,
                         int64_t batch, int width, int act, float* out, int64_t ldo, int32_t* err_flag /*nullable*/,
                         void* stream);
int ctr_act_mask_bwd(float* g, int64_t ldg, const float* y, int64_t ldy, int64_t m, int n, int act, void* stream);

/* ------------------------------------------------------------------------
 * NeuralCF (model/neuralcf.py:33-59) when the vocabularies are much smaller than the batch: the first tower layer and
 * its whole backward are moved from the samples to the table rows.  Linear(cat(MLP_U[u], MLP_I[i])) ==
 * (MLP_U W0a^T)[u] + (MLP_I W0b^T + b0)[i]: two products over num_users + num_items rows instead of one over the
 * batch; backward: segment sums S_U / S_I of ONE 64-float row per sample, then dMLP = S W0half, dW0half = S^T MLP,
 * db0 = colsum S_U; the GMF tables' gradients and the head's GMF weights from T_U[u] = sum_b gz_b GMF_I[i_b]
 * likewise (csrc/ncf_proj.hip has the derivation).  Same function, same gradients as the per-sample path
 * (ctr_embed_mlp_head_fwd / _bwd) up to fp32 summation order.
 *
 * Pattern (anything else: CTR_ELIMIT, nothing enqueued -- use the per-sample entry points): mlp_dim == mf_dim == 64,
 * layers = {128->64, 64->32, 32->16, 16->8} all ReLU with biases, proj (`linear`) 8 -> 64, head (`linear2`) on
 * [gmf | linear(h)] (128 weights), num_users + num_items <= CTR_NCF_PROJ_MAX_ROWS, 16-byte aligned tables.
 * layers[0].y is not used (the first layer's output never exists per sample); layers[1..3].y are the saved
 * activations (m + 1, 32), (m + 1, 16), (m + 1, 8) the forward writes and the backward reads.  EVERY per-sample
 * buffer has one spare row behind the batch (prob: m + 1 elements; ranks: 2 (m + 1) int32): lanes without a sample
 * store / read there, so that no store of the per-sample kernels is conditional.
 * Caller-owned buffers the forward fills for the backward: ptab (num_users + num_items, 64), wfold (76 floats), and
 * with training != 0: ranks and counts ((num_users + num_items) * CTR_NCF_PROJ_COUNT_STRIDE int32: a row's counter
 * has a 64-byte line to itself, same-line atomics are served one after the other).  counts must be ALL ZERO when a training forward
 * is enqueued (its first launch takes every sample's rank with a returning atomic on them); ctr_ncf_proj_bwd's last
 * launch leaves them all zero again, so a caller that pairs every training forward with its backward zero-fills the
 * buffer once.  Parameters must not change between the forward and the backward (the backward re-reads tables and
 * ptab). */
#define CTR_NCF_PROJ_MAX_ROWS 16384
#define CTR_NCF_PROJ_COUNT_STRIDE 16   /* int32 between the counters of two table rows: one 64-byte line each */
typedef struct ctr_ncf_proj {
  const int64_t* user_idx; int64_t user_stride;   /* ids of the batch, element strides */
  const int64_t* item_idx; int64_t item_stride;
  int64_t batch, num_users, num_items;
  const float* mlp_user; const float* mlp_item;   /* (num, mlp_dim) MLP_Embedding_User / _Item */
  const float* gmf_user; const float* gmf_item;   /* (num, mf_dim)  GMF_Embedding_User / _Item */
  int32_t mlp_dim, mf_dim;
  ctr_mlp_layer_t layers[4];                      /* dnn_network (gw / gb unused here) */
  const float* proj_w; int64_t ld_proj_w; const float* proj_b; int32_t proj_n, proj_k;   /* `linear` (proj_n, proj_k) */
  const float* head_w; const float* head_b; int32_t head_act;                            /* `linear2` (1, 2 mf_dim) */
  float* prob; int64_t ldprob;                    /* (batch) output */
  int32_t* err_flag;                              /* nullable */
  float* ptab; float* wfold; int32_t* counts; int32_t* ranks;
  int32_t training;
  int32_t phases;                                 /* 0: the whole call; else a mask of its launches (forward: 1 projected tables
                                                     + head fold, 2 per-sample kernel) -- for per-kernel timing */
} ctr_ncf_proj_t;
typedef struct ctr_ncf_proj_grad {
  const float* gprob; int64_t ldgprob;
  ctr_mlp_layer_t layers[4];                      /* gw / gb of dnn_network (+=); other members unused */
  float* g_mlp_user; float* g_mlp_item; float* g_gmf_user; float* g_gmf_item;   /* dense table gradients (+=) */
  float* g_proj_w; int64_t ld_g_proj_w; float* g_proj_b; float* g_head_w; float* g_head_b;   /* (+=) */
  float* workspace; int64_t workspace_floats;     /* >= ctr_ncf_proj_workspace_floats(batch, users, items) */
  float* zero_buf; int64_t zero_floats;           /* nullable: cleared by the call's first launch (see ctr_embed_mlp_head_bwd) */
  int32_t phases; int32_t reserved;               /* 0: the whole call; else a mask (1 per-sample kernel, 2 segment sums + slab
                                                     reduction + head fold, 4 table-row products), issued in that order */
} ctr_ncf_proj_grad_t;
int ctr_ncf_proj_workspace_floats(int64_t batch, int64_t num_users, int64_t num_items, int64_t* floats /*host, out*/);
int ctr_ncf_proj_fwd(const ctr_ncf_proj_t* d, void* stream);
int ctr_ncf_proj_bwd(const ctr_ncf_proj_t* d, const ctr_ncf_proj_grad_t* g, void* stream);

/* gy: gradient of the LAST layer's output; gx (nullable): gradient of x.  workspace is
 * required: (number of workgroups <= 256) * sum_i (n_i*k_i + n_i) floats. */
int ctr_mlp_bwd(const float* x, int64_t ldx, int64_t m, const ctr_mlp_layer_t* layers, int nlayers,
                const float* gy, int64_t ldgy, float* gx, int64_t ldgx,
                float* workspace, int64_t workspace_floats, void* stream);

/* ------------------------------------------------------------------------
 * Row-sharded tables across the GPUs of a node (no counterpart in the reference,
 * which is single-device): owner(row) = row % world, local
 * row = row / world.  Buckets `n` global ids by owner ahead of the RCCL all-to-all:
 *   counts[w]  = ids owned by rank w, w < world; counts[world] = ids outside [0, vocab)  (world + 1 int64, written)
 *   send[slot] = local row as int32 (the wire format), buckets back to back in rank order
 *   perm[i]    = slot of id i;  inv[slot] = i           (order inside a bucket is not fixed)
 * cursor: CTR_SHARD_SCRATCH_INT64(world) int64 of scratch (per-workgroup histograms; no global atomics are used).
 * Ids outside [0, vocab) travel as row 0 and are counted.
 * ---------------------------------------------------------------------- */
#define CTR_SHARD_SCRATCH_INT64(world) (256 * ((world) + 1))
int ctr_shard_bucket(const int64_t* ids, int64_t n, int world, int64_t vocab, int64_t* counts, int64_t* cursor,
                     int32_t* send, int64_t* perm, int64_t* inv, void* stream);
/* The capacity-bounded layout of the same bucketing: bucket w is slots [w*cap, (w+1)*cap) of `send` / `inv`
 * (world*cap entries each) whatever the counts are, so the all-to-all has equal, host-known splits and a fresh id
 * tensor needs no host read before its ids travel.  send[slot] = local row or -1 (unused); perm[i] = slot of id i;
 * inv[slot] = i (an unused slot names some id of the batch: its row is masked out by the owner).  cursor:
 * CTR_SHARD_SCRATCH_INT64(world) int64 of scratch.
 * state (4 int64, written): {1 if a bucket received more than cap ids -- those ids are NOT placed, the caller must
 * fall back to ctr_shard_bucket --, ids outside [0, vocab), n, -n}: MAX-all-reduced by the caller. */
int ctr_shard_bucket_padded(const int64_t* ids, int64_t n, int world, int64_t vocab, int64_t cap, int64_t* cursor,
                            int32_t* send, int64_t* perm, int64_t* inv, int64_t* state, void* stream);
/* owner side of the padded exchange: rows[j] = recv[j] if it is a row of this shard, else j % local_rows (a harmless
 * target: the slot's gradient row is multiplied by valid[j] = 0); valid[j] = 1.0 / 0.0; mark[j] = recv[j] or -1 (what
 * ctr_rows_mark skips) */
int ctr_shard_recv_rows(const int32_t* recv, int64_t slots, int64_t local_rows, int64_t* rows, float* valid,
                        int64_t* mark, void* stream);
/* table[idx[i]][0:dim] = 0 for i < n (idx outside [0, rows) skipped): clears the rows one step touched in a
 * persistent dense shard-gradient buffer instead of zero-filling the whole shard */
int ctr_rows_zero(float* table, int64_t ld, int64_t rows, int dim, const int64_t* idx, int64_t n, void* stream);

/* Gradient of two (V, 1) first-order tables indexed by the float id columns `user_col` / `item_col` of the (B, ldx)
 * feature matrix (model/ffm.py:19-26 `user` / `item` embeddings, lr.py / widedeep.py / deepfm.py first-order terms):
 * guser1[x[b, user_col]] += v_b, gitem1[x[b, item_col]] += v_b with v_b = g[b*ldg], or g[b*ldg] * p (1 - p) with
 * p = prob[b*ldp] when prob is not NULL (the dlogit of a sigmoid head).  Ids outside their table are skipped.  Either
 * gradient may be NULL.  Small tables only (num_users + num_items <= CTR_ROWS1_MAX_ROWS: they are summed in LDS per
 * workgroup; CTR_ELIMIT beyond -- the interaction kernels' own first-order atomics are the path for large tables). */
#define CTR_ROWS1_MAX_ROWS 32768
int ctr_rows1_scatter(const float* x, int64_t ldx, int user_col, int item_col, const float* g, int64_t ldg,
                      const float* prob, int64_t ldp, int64_t batch, float* guser1, int64_t num_users, float* gitem1,
                      int64_t num_items, void* stream);

/* ------------------------------------------------------------------------
 * Ranking step of recommendation() (model/mf.py:28-35, neuralcf.py:61-72, pnn.py:133-143, din.py:55-66: torch.topk
 * over one user's candidate scores): for each of `rows` rows of n scores (element (r, j) at
 * scores[r*row_stride + j*col_stride]) the indices of the k best, best first.  Order: score descending, NaN first
 * (torch's convention), equal scores by ascending index (fixed, where torch leaves ties unspecified).
 * idx_out (rows, k) int64; val_out (rows, k) nullable.  k <= 4096 (CTR_ELIMIT beyond), k <= n < 2^31.
 * ---------------------------------------------------------------------- */
#define CTR_TOPK_MAX_K 4096
int ctr_topk_rows(const float* scores, int64_t row_stride, int64_t col_stride, int64_t rows, int64_t n, int k,
                  int64_t* idx_out, float* val_out, void* stream);

/* ------------------------------------------------------------------------
 * The host work either side of the step, on the device.
 * ctr_negative_sample (sampler/sampler.py:16-48): for every user u < num_users, num_negatives items drawn
 * uniformly from [0, num_items), each redrawn while bit `item` of the user's row of `excluded`
 * (num_users x words_per_user uint32, bit i of word i/32 = pair (u, i) observed) is set.
 * users[u*num_negatives + j] = u, items[...] = the draw; counter-based generator: the sample depends on
 * `seed` only.  *fail_flag (nullable) is raised if a slot found no free item in 16384 draws.
 * ctr_assemble_features (data/reader.py:98-101): out[b, :] = [users[b], items[b] as floats,
 * user_feat[users[b], :user_width], item_feat[items[b], :item_width]] -- the (B, 45) matrix of the feature
 * models at user_width = 24 (age, 2 gender, 21 occupation columns), item_width = 19 (genre flags).
 * ---------------------------------------------------------------------- */
int ctr_negative_sample(const uint32_t* excluded, int64_t words_per_user, int64_t num_users, int64_t num_items,
                        int num_negatives, uint64_t seed, int64_t* users, int64_t* items, int32_t* fail_flag,
                        void* stream);
int ctr_assemble_features(const int64_t* users, const int64_t* items, int64_t n, const float* user_feat,
                          int user_width, int64_t num_users, const float* item_feat, int item_width,
                          int64_t num_items, float* out, int64_t ldo, int32_t* err_flag, void* stream);

/* ------------------------------------------------------------------------
 * Head folding: a linear layer W (n x k, bias b) whose output feeds ONLY a single-unit layer u is the
 * k-wide dot product  (h W^T + b).u + b2 == h.v + c,  v = W^T u,  c = b.u + b2.  NeuralCF ends like
 * that (model/neuralcf.py:27 linear, :50-56 cat + linear2): folding per step keeps the 8 -> mf_dim
 * layer, its (B, mf_dim) output and that output's gradient out of the batch-sized work.
 *   u_full = [ p columns that pass through (the GMF half of linear2.weight) | n columns fed by W ]
 * fwd: wfold[0:p] = u_full[0:p], wfold[p:p+k] = W^T u, cfold[0] = b.u + b2[0]   (b, b2 may be null)
 * bwd: from gwfold (p+k) = d/d wfold and gc[0] = d/d cfold, all outputs ACCUMULATED (nullable):
 *      gu_full += [gwfold[0:p] | W gwfold[p:] + b gc],  gw += u (x) gwfold[p:],  gb += u gc,  gb2 += gc.
 * ---------------------------------------------------------------------- */
int ctr_fold_head_fwd(const float* u_full, int p, const float* w, int64_t ldw, const float* b, const float* b2, int n,
                      int k, float* wfold, float* cfold, void* stream);
int ctr_fold_head_bwd(const float* u_full, int p, const float* w, int64_t ldw, const float* b, int n, int k,
                      const float* gwfold, const float* gc, float* gu_full, float* gw, int64_t ldgw, float* gb,
                      float* gb2, void* stream);

/* ------------------------------------------------------------------------
 * The rest of a train_loop body (trainer/trainer.py:37-39).
 * ---------------------------------------------------------------------- */

/* torch.nn.BCELoss() with mean reduction (every script, e.g. scripts/pnn.py:54):
 * loss[0] = mean_i -[t_i*max(log p_i,-100) + (1-t_i)*max(log(1-p_i),-100)].
 * prob/target: element i at [i*ld]; workspace: >= 256 floats of device scratch; ticket: one device
 * word that is ZERO on entry and zero again when the call has run (the single launch finds its
 * last workgroup with it) -- allocate it once per stream, zeroed, and keep passing it.
 * gprob_unit (nullable, n contiguous floats): receives what ctr_bce_bwd would write for gloss[0] == 1
 * (bit-identical), so that a caller who knows its upstream gradient is 1 -- loss.backward() -- needs no
 * second launch. */
int ctr_bce_fwd(const float* prob, int64_t ldp, const float* target, int64_t ldt, int64_t n, float* loss,
                float* workspace, int64_t workspace_floats, unsigned int* ticket, float* gprob_unit, void* stream);
/* gprob[i*ldg] = (p_i - t_i) / max(p_i (1-p_i), 1e-12) * gloss[0] / n */
int ctr_bce_bwd(const float* prob, int64_t ldp, const float* target, int64_t ldt, int64_t n,
                const float* gloss, float* gprob, int64_t ldg, void* stream);

/* history-position part of the DIN / DIEN table gradient for the E-wide operand (CTR_DIN_H):
 *   gtable[hist[b,l], :] += gh[(b*len+l)*ldgh + 0:E] + attn[b*len+l] * gpool[(summed ? b : b*len+l)*ldgp + 0:E]
 * gh = gradient of the attention MLP's E-wide input, gpool = gradient of the pooled output (model/din.py:47) or of
 * the per-position outputs (model/dien.py:37).  fp32 atomics, row 0 pre-reduced per workgroup.  dim a power of two
 * <= 256.  The target rows are scattered separately (ctr_embed_bwd with the target ids). */
int ctr_din_scatter_bwd(const int64_t* hist, int64_t vocab, int64_t batch, int len, int dim, const float* gh,
                        int64_t ldgh, const float* attn, const float* gpool, int64_t ldgp, int summed, float* gtable,
                        void* stream);

/* FFM forward in one launch (model/ffm.py:46-86): x is the (batch, >= 45) feature matrix; tables[12] the
 * field-aware tables in the order age_user, age_item, gender_user, gender_item, occupation_user, occupation_item,
 * movie_user, movie_item, userid_user, userid_item, itemid_user, itemid_item (each (vocab, dim), dim in
 * {8, 16, 32, 64}; the bag tables have 1 / 2 / 21 / 19 rows, model/ffm.py:11-26); user1 / item1 the (vocab, 1)
 * bias tables, lin_w / lin_b the Linear(43, 1).  Writes emb (batch, 12*dim) -- the vectors, identical to what
 * ctr_embed_fwd produces for the same field list, kept for the backward -- and
 *   prob[b*ldp] = sigmoid(user1[u] + item1[i] + sum_c (x[b, 2+c] + cross) lin_w[c] + lin_b),
 *   cross = the 15 dot products of ffm.py:62-80 summed left to right. */
int ctr_ffm_fused_fwd(const float* x, int64_t ldx, int64_t batch, int dim, const float* const* tables,
                      int64_t num_users, int64_t num_items, const float* user1, const float* item1,
                      const float* lin_w, const float* lin_b, float* emb, int64_t lde, float* prob, int64_t ldp,
                      int32_t* err_flag, void* stream);

/* backward of ctr_ffm_fused_fwd's head and dot products from the emb it wrote: dz = gprob * prob (1 - prob);
 * guser1[u] += dz, gitem1[i] += dz (fp32 atomics), glin_b += sum dz, glin_w[c] += sum dz (x[b,2+c] + cross)
 * (fixed-order partials through the workspace, >= 1024 * 44 floats), and
 * gemb[b, f*dim : (f+1)*dim] = dz * sum(lin_w) * (sum of the vectors paired with f in ffm.py:62-80), which
 * ctr_embed_bwd then turns into the table gradients.  Any gradient pointer may be NULL. */
int ctr_ffm_fused_bwd(const float* x, int64_t ldx, int64_t batch, int dim, const float* emb, int64_t lde,
                      int64_t num_users, int64_t num_items, const float* lin_w, const float* prob, int64_t ldp,
                      const float* gprob, int64_t ldgp, float* guser1, float* gitem1, float* glin_w, float* glin_b,
                      float* gemb, int64_t ldg, float* workspace, int64_t workspace_floats, void* stream);

/* DIN attention on the E-wide operand (model/din.py:39-44: W1 [h, h-t, t] = (Wa+Wb) h + (Wc-Wb) t, so the
 * first attention layer over all B*L positions only contracts the E columns of h; the per-sample term
 * u[b] = (Wc-Wb) t_b + b1 is added per GROUP of L consecutive rows):
 *   y[i, :] = act(x[i, :] W^T + bias + res[i / group, :])       (bias nullable; n <= 128)
 * mask (nullable; n % 32 == 0, ldmask >= n/32 words): bit (j & 31) of mask[i*ldmask + j/32] = (y[i, j] > 0), the
 * ReLU derivative for ctr_linear_dx_masked -- 1 bit instead of a 4-byte re-read of y per element in backward. */
int ctr_linear_group_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias /*nullable*/,
                         const float* res, int64_t ldr, int group, float* y, int64_t ldy,
                         uint32_t* mask /*nullable*/, int64_t ldmask, int64_t m, int n, int k, int act, void* stream);
/* input gradient of a layer with the activation mask of the layer BELOW folded into the write-back, plus the
 * per-group column sums the grouped term above needs in backward:
 *   gx[i, :] = ((gy[i, :] * act'(y[i, :])) W) * act_in'(xin[i, :])      (xin = this layer's input = the
 *                                                                         previous layer's activation output)
 *   gsum[i / group, :] += gx[i, :]                                      (nullable; fp32 atomics; group >= 32)
 * xmask (nullable; act_in = RELU, k % 32 == 0): the sign bits ctr_linear_group_fwd wrote for xin, used instead of
 * reading xin.  k <= 128. */
int ctr_linear_dx_masked(const float* w, int64_t ldw, const float* y, int64_t ldy, const float* gy, int64_t ldgy,
                         int act, const float* xin /*nullable with xmask*/, int64_t ldxin, int act_in,
                         const uint32_t* xmask /*nullable*/, int64_t ldxmask, float* gx, int64_t ldgx,
                         float* gsum /*nullable*/, int64_t ldgsum, int group, int64_t m, int n, int k, void* stream);
/* a layer and the single-unit layer on top of it in one pass (DIN attention layers 2 and 3, model/din.py:45-46):
 *   y = act(x W^T + bias)   (m, n) stored as usual,     out[i*ldout] = y[i, :] . u + c[0]     (c nullable)
 * n <= 128: a workgroup holds whole output rows and reduces them in its epilogue. */
int ctr_linear_fwd_dot(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias /*nullable*/,
                       float* y, int64_t ldy, const float* u, const float* c /*nullable*/, float* out, int64_t ldout,
                       int64_t m, int n, int k, int act, void* stream);
/* input gradient of DIN's first attention layer on the E-wide operand, scattered straight into the item table's
 * gradient (model/din.py:35-44 backward: the history rows h = table[hist] receive gX plus the pooling's share):
 *   table[idx[i], :] += gy[i, :] W + attn[i] * gpool[i / group, :]          i < m;  W is (n, k), table (vocab, k)
 * The (m, k) gradient itself is never stored.  fp32 atomics; rows with idx 0 (the padding id, a quarter of DIN's
 * positions) are summed per workgroup first; ids outside [0, vocab) add nothing.  k % 32 == 0, k <= 128,
 * group >= 32, vocab < 2^31. */
int ctr_linear_dx_scatter(const float* w, int64_t ldw, const float* gy, int64_t ldgy, const int64_t* idx,
                          const float* attn, const float* gpool, int64_t ldgp, int group, float* table,
                          int64_t vocab, int64_t m, int n, int k, void* stream);
/* backward of the single-unit score layer y = x w^T + b (model/din.py:46, Linear(.., 1)) with the derivative of
 * the activation that produced x folded in, so the gradient that leaves is already the pre-activation gradient
 * of the layer below:
 *   gx = (gy w) * act_in'(x)   (gx may be x itself -- x is dead after this in DIN's backward)
 *   gw += gy^T x,  gb += sum gy        (either nullable; fixed-order partials through the workspace) */
int ctr_linear_n1_bwd_masked(const float* x, int64_t ldx, const float* w, const float* gy, int64_t ldgy, int act_in,
                             float* gx, int64_t ldgx, float* gw /*nullable*/, float* gb /*nullable*/, int64_t m, int k,
                             float* workspace, int64_t workspace_floats, void* stream);

/* ---- opt-in sparse mode of the embedding gradient / optimizer (replaces, for the rows a
 * batch touches, what `optim.Adam(model.parameters(), lr, weight_decay=1e-5)` does to whole tables:
 * scripts/din.py:87, trainer/trainer.py:39).  A table in sparse mode owns persistent device state: a
 * (vocab, dim) gradient accumulation buffer that is all-zero outside pending rows, one int32 flag per row
 * (0 = clean), an int32 list of pending rows (capacity vocab) with its length, and a ticket word. */
typedef struct ctr_rows_mark_t {
  const void* ids;       /* the ids the backward just scattered: int64, or float32 (ids_are_float) */
  int64_t stride;        /* elements between consecutive ids (a column of a (B,F) matrix: F) */
  int64_t n;             /* number of ids */
  int64_t vocab;         /* rows of the table (< 2^31); ids outside [0, vocab) are skipped */
  int32_t* flags;        /* [vocab] */
  int32_t* rows;         /* [vocab] pending-row list */
  int32_t* count;        /* [1] its length */
  int32_t ids_are_float;
  int32_t reserved;
} ctr_rows_mark_t;

typedef struct ctr_rows_table_t {
  float* param;          /* (vocab, dim); NULL allowed for ctr_rows_discard */
  float* grad;           /* (vocab, dim) persistent accumulation buffer */
  float* exp_avg;
  float* exp_avg_sq;
  int32_t* flags;
  int32_t* rows;
  int32_t* count;
  uint32_t* done;        /* [1] zero-initialised ticket */
  int32_t dim;
  int32_t reserved;
} ctr_rows_table_t;

/* after the scatter kernels have added a batch's gradient rows into `grad`: append every row touched for the
 * first time since the last ctr_adam_rows / ctr_rows_discard to the table's pending list.  One job per
 * (table, id list); several jobs may name the same table.  njobs <= CTR_MAX_FIELDS. */
int ctr_rows_mark(const ctr_rows_mark_t* jobs, int njobs, void* stream);
/* Adam on the pending rows only (same update rule as ctr_adam_step, L2 added to the gradient of those rows),
 * then grad rows, flags and the list are left clean.  `step` = this parameter's 1-based step count. */
int ctr_adam_rows(const ctr_rows_table_t* tables, int ntables, double lr, double beta1, double beta2,
                  double eps, double weight_decay, int64_t step, void* stream);
/* drop the pending gradient rows (what zero_grad means in sparse mode) */
int ctr_rows_discard(const ctr_rows_table_t* tables, int ntables, void* stream);

/* torch.optim.Adam(params, lr, betas, eps, weight_decay) (e.g. scripts/pnn.py:55), one
 * step over all tensors in one launch (16-byte aligned, contiguous fp32):
 *   g += wd*p; m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g;
 *   p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps).   tensors: host array. */
#define CTR_ADAM_MAX_TENSORS 64
typedef struct ctr_adam_tensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t numel;
} ctr_adam_tensor_t;
int ctr_adam_step(const ctr_adam_tensor_t* tensors, int ntensors, double lr, double beta1, double beta2,
                  double eps, double weight_decay, int64_t step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CTRHIP_H */
