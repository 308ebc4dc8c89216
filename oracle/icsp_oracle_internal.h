/* icsp_oracle_internal.h — helpers shared by the encoder restatement (icsp_oracle.c) and the decoder restatement
 * (icsp_oracle_dec.c).  TEST INFRASTRUCTURE ONLY, like everything under oracle/. */
#ifndef ICSP_ORACLE_INTERNAL_H
#define ICSP_ORACLE_INTERNAL_H
int icspo_median3(int a, int b, int c);                                   /* the reference's if-chain median (ENC:3677-3679) */
int icspo_luma_dcpred(const int* rec, int r8, int c8, int cols8);         /* ENC:3652-3818 == DEC:2984-3330 */
int icspo_chroma_dcpred(const int* rec, int n, int sw);                   /* ENC:4482-4513 == DEC:4124-4219 */
void icspo_mv_pred(const int* mx, const int* my, int n, int sw, int* px, int* py);   /* ENC:2353-2425 == DEC:4301-4370 */
#endif
