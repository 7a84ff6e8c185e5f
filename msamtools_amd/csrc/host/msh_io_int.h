/* msh_io_int.h -- what the I/O translation units share with each other and with nobody else (msh_io.c: strings, header, aux
 * fields, threads, CRC; msh_sam.c; msh_in.c; msh_out.c). */
#ifndef MSH_IO_INT_H
#define MSH_IO_INT_H
#include "msh.h"
void msh_put_le32(kstr *k, uint32_t v);
void msh_put_le16(kstr *k, uint32_t v);
size_t msh_aux_type_size(int t);
#define MSH_MAX_THREADS 128
#define BGZF_MAX 65536
void msh_hdr_add_target(msh_hdr *h, const char *name, size_t nl, uint32_t len);
void msh_hdr_targets_from_text(msh_hdr *h);
void msh_hdr_forget_names(const msh_hdr *h);      /* the name -> tid table built for this header is let go (msh_close) */
#endif
