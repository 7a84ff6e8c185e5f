/*
 * msh_genome.c -- `profile --genome`: sequence -> genome map and the order of the genomes
 * in the output (msam_profile.c:760-852).
 *
 * The reference collects the genome names in a string hash table and takes the features in
 * the order that table lists its keys (zoeTools.c:202-363): insertion order, except that the
 * table is rebuilt every time it fills up (keys / slots >= 2, slots = 4, 16, 64, ...), and a
 * rebuild re-inserts the keys bucket by bucket -- so after the 8th, 32nd, 128th ... distinct
 * name the order so far becomes "by bucket of the previous table".  msh_keyset reproduces that
 * walk (tests/golden/genome_order_vectors.json holds key orders produced by the reference's
 * own table).  It also serves as the name -> value lookup for the sequence names.
 */
#include <math.h>

#include "msh.h"

struct msh_keyset {
	int32_t n;            /* distinct keys */
	int32_t cap;
	char **key;           /* by id = order of first insertion */
	int32_t *val;
	int32_t *walk;        /* ids in key-walk order */
	int level, slots;
	int32_t *head;        /* per slot: first id of its chain, -1 = empty */
	int32_t *tail;
	int32_t *next;        /* per id: next id in the same slot */
};

/* zoeTools.c:202-228: slot = (int)(slots * frac(sum of key[i] * m[i % 7])), key[i] as (signed) char */
static int keyset_slot(int slots, const char *key) {
	static const double m[7] = {3.1415926536, 2.7182818285, 1.6180339887, 1.7320508076,
	                            2.2360679775, 2.6457513111, 3.3166247904};
	double sum = 0;
	size_t i, len = strlen(key);
	for (i = 0; i < len; i++) sum += key[i] * m[i % 7];
	return (int)(slots * (sum - floor(sum)));
}

static void keyset_link(msh_keyset *k, int32_t id) {
	const int s = keyset_slot(k->slots, k->key[id]);
	k->next[id] = -1;
	if (k->head[s] < 0) k->head[s] = id; else k->next[k->tail[s]] = id;
	k->tail[s] = id;
}

/* zoeTools.c:230-279: next level; the keys are taken slot by slot out of the old table, and
 * that becomes their new order */
static void keyset_grow(msh_keyset *k) {
	const int old_slots = k->slots;
	int32_t *old_head = k->head, *old_next = NULL, *order = NULL, i, n = 0;
	int s;
	if (k->n > 0) {
		old_next = (int32_t *)malloc(sizeof(int32_t) * (size_t)k->n);
		order = (int32_t *)malloc(sizeof(int32_t) * (size_t)k->n);
		memcpy(old_next, k->next, sizeof(int32_t) * (size_t)k->n);
		for (s = 0; s < old_slots; s++)
			for (i = old_head[s]; i >= 0; i = old_next[i]) order[n++] = i;
	}
	k->level++;
	k->slots = (int)pow(4, k->level);
	free(k->tail);
	k->head = (int32_t *)malloc(sizeof(int32_t) * (size_t)k->slots);
	k->tail = (int32_t *)malloc(sizeof(int32_t) * (size_t)k->slots);
	for (s = 0; s < k->slots; s++) k->head[s] = k->tail[s] = -1;
	for (i = 0; i < n; i++) {
		k->walk[i] = order[i];
		keyset_link(k, order[i]);
	}
	free(old_head);
	free(old_next);
	free(order);
}

msh_keyset *msh_keyset_new(void) {
	msh_keyset *k = (msh_keyset *)calloc(1, sizeof *k);
	keyset_grow(k);                     /* level 1: 4 slots (zoeTools.c:304-314) */
	return k;
}

int32_t msh_keyset_find(const msh_keyset *k, const char *key) {
	int32_t i;
	for (i = k->head[keyset_slot(k->slots, key)]; i >= 0; i = k->next[i])
		if (strcmp(k->key[i], key) == 0) return i;
	return -1;
}

/* zoeSetHash (zoeTools.c:330-357): an existing key only gets the new value */
int32_t msh_keyset_put(msh_keyset *k, const char *key, int32_t val) {
	int32_t id = msh_keyset_find(k, key);
	if (id >= 0) { k->val[id] = val; return id; }
	if (k->n == k->cap) {
		k->cap = k->cap ? 2 * k->cap : 64;
		k->key = (char **)realloc(k->key, sizeof(char *) * (size_t)k->cap);
		k->val = (int32_t *)realloc(k->val, sizeof(int32_t) * (size_t)k->cap);
		k->walk = (int32_t *)realloc(k->walk, sizeof(int32_t) * (size_t)k->cap);
		k->next = (int32_t *)realloc(k->next, sizeof(int32_t) * (size_t)k->cap);
	}
	id = k->n++;
	k->key[id] = strdup(key);
	k->val[id] = val;
	k->walk[id] = id;
	keyset_link(k, id);
	if ((float)k->n / (float)k->slots >= 2.0f) keyset_grow(k);
	return id;
}

int32_t msh_keyset_size(const msh_keyset *k) { return k->n; }
int32_t msh_keyset_walk(const msh_keyset *k, int32_t pos) { return k->walk[pos]; }
const char *msh_keyset_key(const msh_keyset *k, int32_t id) { return k->key[id]; }
int32_t msh_keyset_value(const msh_keyset *k, int32_t id) { return k->val[id]; }

void msh_keyset_free(msh_keyset *k) {
	int32_t i;
	if (!k) return;
	for (i = 0; i < k->n; i++) free(k->key[i]);
	free(k->key); free(k->val); free(k->walk); free(k->next); free(k->head); free(k->tail);
	free(k);
}

/* msam_profile.c:760-852.  Returns fmap[n_targets]; features in the reference's key-walk order. */
int32_t *msh_genome_map(const char *path, const msh_hdr *h, int32_t *n_features, char ***names, uint32_t **lens) {
	FILE *f = fopen(path, "r");
	char line[8192], g[4096], s[4096];
	int32_t *fmap = (int32_t *)malloc(sizeof(int32_t) * (size_t)(h->n_targets ? h->n_targets : 1)), i, nf;
	int32_t *feature_of;        /* genome id -> position in the key walk */
	msh_keyset *seqs = msh_keyset_new(), *genomes = msh_keyset_new();
	if (!f) mDie("Cannot open file %s", path);
	for (i = 0; i < h->n_targets; i++) {
		fmap[i] = -1;
		msh_keyset_put(seqs, h->target_name[i], i);         /* :771-777 (a repeated name keeps its last tid) */
	}
	while (fgets(line, sizeof line, f)) {                   /* :782-795: the genome names */
		if (sscanf(line, "%4095s\t%4095s", g, s) != 2) mDie("GENOME DEFINITION LINE ERROR");
		msh_keyset_put(genomes, g, 1);
	}
	nf = msh_keyset_size(genomes);                          /* :797-805: genome -> index in key order */
	feature_of = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nf ? nf : 1));
	for (i = 0; i < nf; i++) feature_of[msh_keyset_walk(genomes, i)] = i;
	rewind(f);
	while (fgets(line, sizeof line, f)) {                   /* :808-829 */
		int32_t gid, sid;
		if (sscanf(line, "%4095s\t%4095s", g, s) != 2) mDie("GENOME DEFINITION LINE ERROR");
		gid = msh_keyset_find(genomes, g);
		sid = msh_keyset_find(seqs, s);
		if (gid < 0) mDie("Genome '%s' not found in BAM file", g);
		if (sid < 0) mDie("Sequence '%s' not found in BAM file", s);
		fmap[msh_keyset_value(seqs, sid)] = feature_of[gid];
	}
	fclose(f);
	*lens = (uint32_t *)calloc((size_t)(nf ? nf : 1), sizeof(uint32_t));
	for (i = 0; i < h->n_targets; i++) {                    /* :832-842 */
		if (fmap[i] == -1) mDie("Sequence '%s' not found in genome definition", h->target_name[i]);
		(*lens)[fmap[i]] += h->target_len[i];
	}
	*names = (char **)malloc(sizeof(char *) * (size_t)(nf ? nf : 1));
	for (i = 0; i < nf; i++) (*names)[i] = strdup(msh_keyset_key(genomes, msh_keyset_walk(genomes, i)));   /* :845-851 */
	*n_features = nf;
	free(feature_of);
	msh_keyset_free(seqs);
	msh_keyset_free(genomes);
	return fmap;
}
