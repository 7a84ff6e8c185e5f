/*
 * msh_main.c -- the program's entry: command dispatch as msamtools.c:8-49 (filter, profile, coverage, summary, help).
 */
#include "msh_cli.h"

#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

/* ------------------------------------------------------------------------ */
/* MSX_DETACH=1: the device's process left behind in a child                 */
/* ------------------------------------------------------------------------ */
/* A process that has used the HIP runtime takes 0.10-0.16 s to be let go of by the driver AFTER _exit (one thread in state D at
 * __synchronize_srcu: profiles/round5/exit_cost.md) -- a quarter of the wall time of a command on the 100 M-record file, and
 * nothing inside the process shortens it.  With MSX_DETACH=1 the command runs in a child forked before anything touches the
 * runtime (fork after that is not safe); when the child has written and closed everything it reports its exit code through a
 * pipe, points its stdout / stderr at /dev/null -- whoever waits for the end of those streams sees it now -- and only then calls
 * _exit; the parent returns the code at once and the child's teardown runs on, owned by init.  A child that ends without a
 * report (a crash, a signal, exit() on the way through the options) is waited for and its fate is passed on.  Signals sent to the
 * parent go to the child.  Not under a profiler or any other preloaded tool (those initialise the runtime before main). */
static pid_t g_child;
static int g_report_fd = -1;
static void pass_signal(int sig) {
	if (g_child > 0) kill(g_child, sig);
	_exit(128 + sig);
}
static void msh_report_and_release(int rc) {
	unsigned char b = (unsigned char)rc;
	int dn;
	if (g_report_fd < 0) return;
	fflush(stdout);
	fflush(stderr);
	dn = open("/dev/null", O_WRONLY);
	if (dn >= 0) { dup2(dn, 1); dup2(dn, 2); if (dn > 2) close(dn); }
	while (write(g_report_fd, &b, 1) < 0 && errno == EINTR) {}
	close(g_report_fd);
	g_report_fd = -1;
}
static void detach_if_asked(void) {
	const char *e = getenv("MSX_DETACH"), *pre = getenv("LD_PRELOAD");
	int sp[2];
	if (!e || atoi(e) == 0 || (pre && *pre) || getenv("HSA_TOOLS_LIB") || getenv("ROCP_TOOL_LIB") || getenv("MSX_CLEAN_EXIT")) return;
	if (pipe(sp) != 0) return;
	fflush(NULL);
	g_child = fork();
	if (g_child < 0) { close(sp[0]); close(sp[1]); return; }
	if (g_child == 0) {          /* the command runs here */
		close(sp[0]);
		g_report_fd = sp[1];
		msh_exit_hook = msh_report_and_release;
		fcntl(g_report_fd, F_SETFD, FD_CLOEXEC);
		return;
	}
	close(sp[1]);
	{
		static const int sigs[] = {SIGINT, SIGTERM, SIGHUP, SIGQUIT, SIGPIPE};
		unsigned char b = 0;
		ssize_t k;
		int st = 0;
		size_t i;
		for (i = 0; i < sizeof sigs / sizeof sigs[0]; i++) signal(sigs[i], pass_signal);
		/* (the parent holds no end of the command's streams open longer than it must) */
		do k = read(sp[0], &b, 1); while (k < 0 && errno == EINTR);
		if (k == 1) _exit(b);
		while (waitpid(g_child, &st, 0) < 0 && errno == EINTR) {}
		if (WIFSIGNALED(st)) { signal(WTERMSIG(st), SIG_DFL); raise(WTERMSIG(st)); _exit(128 + WTERMSIG(st)); }
		_exit(WIFEXITED(st) ? WEXITSTATUS(st) : 1);
	}
}

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
	detach_if_asked();
	{
		int (*cmd)(int, char **) = strcmp(argv[1], "filter") == 0 ? msam_filter_main : strcmp(argv[1], "profile") == 0 ? msam_profile_main :
		                           strcmp(argv[1], "coverage") == 0 ? msam_coverage_main : strcmp(argv[1], "summary") == 0 ? msam_summary_main : NULL;
		if (cmd) {
			/* (a command that comes back here instead of leaving through _exit has the runtime's exit handlers ahead of it:
			 *  the warm-up thread must not be inside the runtime then -- msh_common.c) */
			const int rc = cmd(argc - 1, argv + 1);
			runtime_warmup_join();
			msh_report_and_release(rc);
			return rc;
		}
	}
	if (strcmp(argv[1], "help") == 0) { usage(stdout); return 0; }
	fprintf(stderr, "[msamtools] unrecognized command '%s'\n", argv[1]);
	usage(stderr);
	return 1;
}
