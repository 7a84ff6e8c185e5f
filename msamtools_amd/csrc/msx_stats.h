// msx_stats.h -- arguments of the per-record statistics + filter kernels (msx_stats.hip, msx_filter.hip)
#ifndef MSX_STATS_H
#define MSX_STATS_H

#include "msx_internal.h"

// pool byte written for k_besthit_select (FilterArgs.pool_as_code): bits 6-7 are FLAG's READ1/READ2
#define MSX_PC_IN 0x01u
#define MSX_PC_HAS_AS 0x02u
#define MSX_PC_UNMAP 0x04u     // FLAG & 4, whether the record is pooled or not: a pool that BEGINS with an unmapped record
                               // continues the QNAME of the pool before it (msam_filter.c:120-125,170; msx_count.h)

struct FilterArgs {
	int64_t n;
	const uint16_t *flag;
	const uint8_t *rflags;
	const uint32_t *cigar_off;
	const uint32_t *cigar;
	const uint32_t *md_off;
	const uint8_t *md;
	const int32_t *nm;
	const int32_t *as;
	int32_t min_length, ppt, max_clip;
	int32_t choice;         // bit0 -l, bit1 -p/--ppt, bit2 -z (msam_filter.c:79-81)
	int32_t rescore, invert, keep_unmapped;
	int32_t wide_ok;        // cigar_off/md_off 8-byte, flag 4-byte, rflags and pool 2-byte aligned
	uint8_t *pool;          // [n] out: 1 = record enters the pool
	int32_t pool_as_code;   // best-hit follows: a pooled record's byte is MSX_PC_IN | MSX_PC_HAS_AS | its mate bits,
	                        // everything k_besthit_select needs to know about it besides its score
	int32_t *as_out;        // [n] out (rescore) or null
	int32_t *o_len, *o_qlen, *o_qclip, *o_edit;   // optional per-record stats
	uint8_t *o_status;
	msx_dev_status *st;
};

// k_aln_stats_flat (msx_stats.hip): one wave per 128 records, MD walked flat
void msx_launch_aln_stats_flat(msx_ctx *ctx, const FilterArgs &A, int grid);

#endif
