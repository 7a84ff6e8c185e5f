/*
 * msh_main.c -- the program's entry: command dispatch as msamtools.c:8-49 (filter, profile, coverage, summary, help).
 */
#include "msh_cli.h"

/* ------------------------------------------------------------------------ */
/* msamtools.c:8-49                                                           */
/* ------------------------------------------------------------------------ */
int usage(FILE *out) {
	fprintf(out, "\n");
	fprintf(out, "Program: %s (Metagenomics-related extension to samtools; MI355X filter/profile path)\n", PROGRAM);
	fprintf(out, "Version: %s (git %s; own BGZF/BAM reader, no htslib)\n", MSH_VERSION, MSH_GIT_COMMIT);
	fprintf(out, "\n");
	fprintf(out, "Usage:   %s <command> [options]\n\n", PROGRAM);
	fprintf(out, "Commands:\n");
	fprintf(out, " -- Filtering\n");
	fprintf(out, "     filter         filter alignments based on alignment statistics\n");
	fprintf(out, "\n");
	fprintf(out, " -- Profiling\n");
	fprintf(out, "     profile        estimate relative abundance profile of reference sequences or genomes in bam file\n");
	fprintf(out, "\n");
	fprintf(out, " -- Coverage\n");
	fprintf(out, "     coverage       estimate per-base or per-sequence read coverage of each reference sequence\n");
	fprintf(out, "\n");
	fprintf(out, " -- Summary\n");
	fprintf(out, "     summary        summarize alignment statistics per read in a table format\n");
	fprintf(out, "\n");
	return 1;
}

int main(int argc, char *argv[]) {
	g_t_main = now_s();
	msh_main_thread = pthread_self();
	msh_main_thread_set = 1;
	if (argc < 2) return usage(stderr);
	{
		int (*cmd)(int, char **) = strcmp(argv[1], "filter") == 0 ? msam_filter_main : strcmp(argv[1], "profile") == 0 ? msam_profile_main :
		                           strcmp(argv[1], "coverage") == 0 ? msam_coverage_main : strcmp(argv[1], "summary") == 0 ? msam_summary_main : NULL;
		if (cmd) {
			/* (a command that comes back here instead of leaving through _exit has the runtime's exit handlers ahead of it:
			 *  the warm-up thread must not be inside the runtime then -- msh_common.c) */
			const int rc = cmd(argc - 1, argv + 1);
			runtime_warmup_join();
			return rc;
		}
	}
	if (strcmp(argv[1], "help") == 0) { usage(stdout); return 0; }
	fprintf(stderr, "[msamtools] unrecognized command '%s'\n", argv[1]);
	usage(stderr);
	return 1;
}
