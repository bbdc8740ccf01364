/* host_fuzz - sanitizer driver for the native host stage (csrc/nrv_host.c + csrc/nrv_host_fast5.c).
 *
 * The native reader parses UNTRUSTED fast5 files as threads of the GPU worker (cli.run_workers); the reference leaves that
 * to h5py / libhdf5 (nanorevutils/nanorev_fast5_handeler.py:39-150).  This driver is built by scripts/host_sanitize.sh with
 *   gcc -fsanitize=address,undefined -fno-sanitize-recover=all      (modes "fuzz", "api")
 *   gcc -fsanitize=thread                                           (mode "threads")
 * and run by tests/test_hostlib_sanitize.py; it INCLUDES the two sources, so the file image enters load_fast5_image
 * directly and the statics (read_dset, walk_chunks ...) locate what is worth mutating.  Any sanitizer report aborts; every
 * outcome of a mutated file must be a return code.
 *
 *   host_fuzz fuzz N SEED TMPDIR file.fast5 ...     N mutations per file and class mix below
 *   host_fuzz api TMPDIR file.fast5 ...             bundle + finisher entry points, argument edge cases
 *   host_fuzz threads NTHREADS ITERS TMPDIR file.fast5 ...
 */
#include "../../nanoreviser_amd/csrc/nrv_host.c"
#include "../../nanoreviser_amd/csrc/nrv_host_fast5.c"

#include <errno.h>
#include <sys/stat.h>

static const char* G = "Basecall_1D_000";
static const char* SG = "BaseCalled_template";

/* ---- xorshift64* ---------------------------------------------------------------------------------------------------- */
static uint64_t rng_s;
static uint64_t rnd(void) {
  rng_s ^= rng_s >> 12; rng_s ^= rng_s << 25; rng_s ^= rng_s >> 27;
  return rng_s * 0x2545F4914F6CDD1Dull;
}
static uint64_t below(uint64_t n) { return n ? rnd() % n : 0; }

static uint8_t* slurp(const char* path, long* n) {
  FILE* f = fopen(path, "rb");
  if (!f) return 0;
  fseek(f, 0, SEEK_END); *n = ftell(f); fseek(f, 0, SEEK_SET);
  uint8_t* d = (uint8_t*)malloc((size_t)*n + 1);
  if (d && fread(d, 1, (size_t)*n, f) != (size_t)*n) { free(d); d = 0; }
  fclose(f);
  return d;
}
static int spit(const char* path, const uint8_t* d, long n) {
  FILE* f = fopen(path, "wb");
  if (!f) return -1;
  const int ok = fwrite(d, 1, (size_t)n, f) == (size_t)n;
  return fclose(f) == 0 && ok ? 0 : -1;
}

/* ---- what the file holds: addresses worth hitting --------------------------------------------------------------------- */
typedef struct { uint64_t addr, csize, key_off; } Chunk;           /* key_off: where the B-tree key's size field lives */
typedef struct {
  uint64_t ev_hdr, sg_hdr, fq_hdr;        /* object headers of Events / Signal / Fastq */
  Dset ev, sg;
  Chunk evc[64], sgc[256];
  int n_evc, n_sgc;
  uint64_t sigs[512]; int n_sigs;         /* offsets of TREE / SNOD / HEAP / GCOL signatures */
  long model_state;                       /* offset of the "model_state" member name */
  uint64_t msgs[512]; int n_msgs;         /* offsets of the header messages of the three datasets + their groups */
} Map;

static void chunks_of(Buf* b, const Dset* ds, uint64_t node, Chunk* out, int* n, int cap, int depth) {
  if (node == UNDEF_ADDR || depth > 8 || b->bad) return;
  const uint8_t* sig = P(b, node, 24);
  if (!sig || memcmp(sig, "TREE", 4)) return;
  const int level = (int)U(b, node + 5, 1), used = (int)U(b, node + 6, 2);
  uint64_t p = node + 24;
  for (int i = 0; i < used && !b->bad; ++i, p += 32) {
    const uint64_t csize = U(b, p, 4), child = U(b, p + 24, 8);
    if (level > 0) { chunks_of(b, ds, child, out, n, cap, depth + 1); continue; }
    if (*n < cap) { out[*n].addr = child; out[*n].csize = csize; out[*n].key_off = p; ++*n; }
  }
}

static int map_file(const uint8_t* d, long n, Map* m) {
  memset(m, 0, sizeof *m);
  Buf B = {d, (size_t)n, 0, 0};
  Buf* b = &B;
  const int sver = d[8];
  const uint64_t root = U(b, 24 + (sver == 1 ? 4 : 0) + 32 + 8, 8);
  uint64_t gaddr, raddr, rd0;
  char pth[256];
  snprintf(pth, sizeof pth, "Analyses/%s", G);
  if (path_lookup(b, root, pth, &gaddr)) return -1;
  snprintf(pth, sizeof pth, "%s/Events", SG);
  if (path_lookup(b, gaddr, pth, &m->ev_hdr)) return -1;
  snprintf(pth, sizeof pth, "%s/Fastq", SG);
  if (path_lookup(b, gaddr, pth, &m->fq_hdr)) m->fq_hdr = 0;
  if (path_lookup(b, root, "Raw/Reads", &raddr) || group_lookup(b, raddr, 0, &rd0) || group_lookup(b, rd0, "Signal", &m->sg_hdr)) return -1;
  if (read_dset(b, m->ev_hdr, &m->ev) || read_dset(b, m->sg_hdr, &m->sg)) return -1;
  if (m->ev.layout == 2) chunks_of(b, &m->ev, m->ev.btree, m->evc, &m->n_evc, 64, 0);
  if (m->sg.layout == 2) chunks_of(b, &m->sg, m->sg.btree, m->sgc, &m->n_sgc, 256, 0);
  const uint64_t hdrs[6] = {m->ev_hdr, m->sg_hdr, m->fq_hdr, gaddr, raddr, rd0};
  for (int h = 0; h < 6; ++h) {
    Obj o;
    if (!hdrs[h] || read_obj(b, hdrs[h], &o)) continue;
    if (m->n_msgs < 512) m->msgs[m->n_msgs++] = hdrs[h];
    for (int i = 0; i < o.n && m->n_msgs < 512; ++i) m->msgs[m->n_msgs++] = o.m[i].off;
  }
  for (long i = 0; i + 4 <= n && m->n_sigs < 512; ++i)
    if (!memcmp(d + i, "TREE", 4) || !memcmp(d + i, "SNOD", 4) || !memcmp(d + i, "HEAP", 4) || !memcmp(d + i, "GCOL", 4))
      m->sigs[m->n_sigs++] = (uint64_t)i;
  m->model_state = -1;
  for (long i = 0; i + 11 <= n; ++i) if (!memcmp(d + i, "model_state", 11)) { m->model_state = i; break; }
  return 0;
}

static const uint64_t kExtreme[] = {0, 1, 2, 0x7f, 0x80, 0xff, 0x7fff, 0x8000, 0xffff, 0x7fffffffull, 0x80000000ull, 0xffffffffull,
                                    0xfffffff0ull, 0x7fffffffffffffffull, 0x8000000000000000ull, 0xffffffffffffffffull,
                                    0x8000000000000c04ull, (uint64_t)1 << 40, (uint64_t)1 << 62, 0xfffffffffffffff8ull};
static uint64_t extreme(void) { return kExtreme[below(sizeof kExtreme / sizeof kExtreme[0])]; }
static void put(uint8_t* d, long n, uint64_t off, uint64_t v, int nb) {
  for (int i = 0; i < nb; ++i) if ((long)(off + (uint64_t)i) < n) d[off + (uint64_t)i] = (uint8_t)(v >> (8 * i));
}

/* rows of the Events table: through the chunk when it is stored plain, re-deflated into place when it is compressed */
static void mutate_events(uint8_t* d, long n, const Map* m) {
  const int es = m->ev.esize;
  const Member *ms = 0, *mm = 0;
  for (int i = 0; i < m->ev.nmem; ++i) {
    if (!strcmp(m->ev.mem[i].name, "start")) ms = &m->ev.mem[i];
    if (!strcmp(m->ev.mem[i].name, "move")) mm = &m->ev.mem[i];
  }
  if (!ms || !mm || es <= 0) return;
  uint8_t* rows = 0;
  uint64_t nrows = 0, at = 0, cap = 0, key_off = 0;
  int deflated = 0;
  if (m->ev.layout == 1) { at = m->ev.addr; nrows = m->ev.dims[0]; cap = nrows * (uint64_t)es; }
  else if (m->ev.layout == 2 && m->n_evc > 0) {
    const Chunk* c = &m->evc[below((uint64_t)m->n_evc)];
    at = c->addr; cap = c->csize; key_off = c->key_off;
    int has_deflate = 0, other = 0;
    for (int f = 0; f < m->ev.nfilt; ++f) { if (m->ev.filt[f] == 1) has_deflate = 1; else other = 1; }
    if (other) return;
    deflated = has_deflate;
    nrows = m->ev.cdim;
  } else return;
  if (at + cap > (uint64_t)n) return;
  if (deflated) {
    uLongf dl = (uLongf)(nrows * (uint64_t)es);
    rows = (uint8_t*)malloc(dl + 1);
    if (!rows || uncompress(rows, &dl, d + at, (uLong)cap) != Z_OK) { free(rows); return; }
    nrows = dl / (uint64_t)es;
  } else { rows = d + at; nrows = cap / (uint64_t)es; }
  if (nrows) {
    const int k = 1 + (int)below(4);
    for (int j = 0; j < k; ++j) {
      uint64_t r = below(nrows);
      for (int tries = 0; tries < 8; ++tries) {          /* a row that counts: move != 0 */
        uint64_t mv = 0;
        memcpy(&mv, rows + r * es + mm->off, (size_t)(mm->t.size < 8 ? mm->t.size : 8));
        if (mv) break;
        r = below(nrows);
      }
      static const uint64_t kStart[] = {0x8000000000000000ull, 0x8000000000000c04ull, 0x7fffffffffffffffull, 0x7ffffffffffffffeull,
                                        0xffffffffffffffffull, 0xffffffffull, 0x80000000ull, (uint64_t)1 << 40, ((uint64_t)1 << 40) + 1,
                                        0xffffff0000000000ull, 0xfffffeffffffffffull, (uint64_t)1 << 62};
      switch (below(4)) {
        case 0: put(rows, (long)(nrows * es), r * es + ms->off, kStart[below(sizeof kStart / sizeof kStart[0])], ms->t.size); break;
        case 1: put(rows, (long)(nrows * es), r * es + mm->off, extreme(), mm->t.size); break;
        case 2: put(rows, (long)(nrows * es), r * es + mm->off, below(4), mm->t.size); break;
        default: put(rows, (long)(nrows * es), r * es + ms->off, rnd() >> below(64), ms->t.size); break;
      }
    }
    if (below(4) == 0)                                   /* every move 0 but a few: "too much zero moves" */
      for (uint64_t r = 0; r < nrows; ++r) if (below(64)) put(rows, (long)(nrows * es), r * es + mm->off, 0, mm->t.size);
  }
  if (deflated) {
    uLongf cl = compressBound((uLong)(nrows * (uint64_t)es));
    uint8_t* z = (uint8_t*)malloc(cl);
    if (z && compress2(z, &cl, rows, (uLong)(nrows * (uint64_t)es), 9) == Z_OK && cl <= cap) {
      memcpy(d + at, z, cl);
      put(d, n, key_off, cl, 4);
    }
    free(z); free(rows);
  }
}

static void mutate(uint8_t* d, long* pn, const Map* m, int cls) {
  long n = *pn;
  switch (cls) {
    case 0: {                                             /* byte flips / random bytes in the metadata region */
      const int k = 1 + (int)below(32);
      for (int i = 0; i < k; ++i) { const uint64_t p = 8 + below(n < 4096 ? (uint64_t)n - 8 : 4088); d[p] = below(2) ? (uint8_t)rnd() : d[p] ^ 0xFF; }
      break;
    }
    case 1: {                                             /* anywhere in the file */
      const int k = 1 + (int)below(32);
      for (int i = 0; i < k; ++i) d[8 + below((uint64_t)n - 8)] = (uint8_t)rnd();
      break;
    }
    case 2: {                                             /* 2 / 4 / 8-byte extreme values inside header messages */
      const int k = 1 + (int)below(3);
      for (int i = 0; i < k && m->n_msgs; ++i) {
        const int nb = 1 << (1 + (int)below(3));
        put(d, n, m->msgs[below((uint64_t)m->n_msgs)] + below(40), extreme(), nb);
      }
      break;
    }
    case 3: {                                             /* B-tree / symbol node / heap: counts, child addresses, cycles */
      const int k = 1 + (int)below(3);
      for (int i = 0; i < k && m->n_sigs; ++i) {
        const uint64_t s = m->sigs[below((uint64_t)m->n_sigs)];
        switch (below(4)) {
          case 0: put(d, n, s + 4 + below(60), extreme(), 1 << (int)below(4)); break;
          case 1: put(d, n, s + 6, extreme(), 2); break;                                  /* entries used */
          case 2: put(d, n, s + 24 + 8 * below(16), below(2) ? s : m->sigs[below((uint64_t)m->n_sigs)], 8); break;   /* a link to itself / to another node */
          default: d[s + below(4)] ^= 0x20; break;
        }
      }
      break;
    }
    case 4: {                                             /* the compound datatype around "model_state" */
      if (m->model_state < 0) break;
      const int k = 1 + (int)below(6);
      for (int i = 0; i < k; ++i) {
        const long p = m->model_state - 300 + (long)below(600);
        if (p >= 8 && p < n) { if (below(2)) d[p] = (uint8_t)rnd(); else put(d, n, (uint64_t)p, extreme(), 4); }
      }
      break;
    }
    case 5: mutate_events(d, n, m); break;               /* rows of the Events table */
    case 6: {                                             /* truncation */
      *pn = 8 + (long)below((uint64_t)n - 8);
      break;
    }
    case 7: {                                             /* chunk keys of Signal / Events: sizes, filter masks, offsets, addresses */
      const int use_ev = m->n_evc && below(3) == 0;
      const Chunk* cs = use_ev ? m->evc : m->sgc;
      const int nc = use_ev ? m->n_evc : m->n_sgc;
      if (!nc) break;
      const Chunk* c = &cs[below((uint64_t)nc)];
      switch (below(5)) {
        case 0: put(d, n, c->key_off, extreme(), 4); break;
        case 1: put(d, n, c->key_off + 4, extreme(), 4); break;
        case 2: put(d, n, c->key_off + 8, extreme(), 8); break;
        case 3: put(d, n, c->key_off + 24, extreme(), 8); break;
        default: if (c->addr + 16 < (uint64_t)n) d[c->addr + below(c->csize < 64 ? c->csize + 1 : 64)] ^= (uint8_t)(1 + below(255)); break;
      }
      break;
    }
    default: break;
  }
}

static long g_codes[5];

/* what a caller does with a read that parsed: the merge + record + file of both formats */
static void finish_some(const nrvh_read* r, const char* tmpdir) {
  const int Ts[3] = {11, 13, 1};
  char dst[600];
  for (int k = 0; k < 3; ++k) {
    const int T = Ts[k];
    const int64_t nwin = r->n_ev - T > 0 ? r->n_ev - T : 0;
    int8_t* a1 = (int8_t*)malloc((size_t)nwin + 1);
    int8_t* a2 = (int8_t*)malloc((size_t)nwin + 1);
    uint8_t* qc = (uint8_t*)malloc((size_t)nwin + 1);
    for (int64_t i = 0; i < nwin; ++i) { a1[i] = (int8_t)below(6); a2[i] = (int8_t)below(5); qc[i] = (uint8_t)(33 + below(60)); }
    int64_t nw = 0;
    snprintf(dst, sizeof dst, "%s/f_out.%s", tmpdir, k & 1 ? "fastq" : "fasta");
    const int rc = nrvh_finish_read(r->bases, r->n_ev, a1, a2, nwin, T, k & 1 ? qc : 0, "a read|||name", dst, k & 1, &nw);
    if (rc != NRVH_OK && rc != NRVH_E_IO) { fprintf(stderr, "finish_read rc %d\n", rc); abort(); }
    free(a1); free(a2); free(qc);
  }
}

static int mode_fuzz(int nmut, uint64_t seed, const char* tmpdir, int nfiles, char** files) {
  char tmpf[600];
  snprintf(tmpf, sizeof tmpf, "%s/mut.fast5", tmpdir);
  long total = 0;
  for (int fi = 0; fi < nfiles; ++fi) {
    long n0;
    uint8_t* orig = slurp(files[fi], &n0);
    if (!orig) { fprintf(stderr, "cannot read %s\n", files[fi]); return 2; }
    Map m;
    if (map_file(orig, n0, &m)) { fprintf(stderr, "cannot map %s\n", files[fi]); return 2; }
    nrvh_read r;
    char err[NRVH_ERR_LEN];
    if (load_fast5_image(orig, n0, G, SG, 1, &r, err, sizeof err) != NRVH_OK) { fprintf(stderr, "the unmutated file does not parse: %s\n", err); return 2; }
    const int64_t n_ev0 = r.n_ev;
    finish_some(&r, tmpdir);
    nrvh_free_read(&r);
    uint8_t* d = (uint8_t*)malloc((size_t)n0);
    rng_s = seed * 0x9E3779B97F4A7C15ull + (uint64_t)fi * 1315423911ull + 1;
    long per_class[8] = {0}, ok_class[8] = {0};
    for (int it = 0; it < nmut; ++it) {
      memcpy(d, orig, (size_t)n0);
      long n = n0;
      const int cls = it % 8;
      mutate(d, &n, &m, cls);
      if (below(8) == 0) mutate(d, &n, &m, (int)below(8));             /* sometimes two kinds at once */
      /* an exact-size copy: an out-of-bounds read of the image is a heap overflow ASan sees */
      uint8_t* img = (uint8_t*)malloc((size_t)n);
      memcpy(img, d, (size_t)n);
      int rc;
      if (it % 16 == 15) {                                             /* through the file + bundle entry points */
        if (spit(tmpf, img, n)) { fprintf(stderr, "cannot write %s\n", tmpf); return 2; }
        const char* paths[3] = {files[fi], tmpf, "/nonexistent/x.fast5"};
        nrvh_bundle b;
        if (nrvh_load_bundle(paths, 3, G, SG, (int)below(2), &b) != NRVH_OK) { fprintf(stderr, "bundle failed\n"); abort(); }
        if (b.status[0] != NRVH_OK || b.status[2] != NRVH_E_IO) { fprintf(stderr, "bundle statuses %d %d\n", b.status[0], b.status[2]); abort(); }
        rc = b.status[1];
        if (rc == NRVH_OK && b.n_ev != n_ev0 + (int64_t)b.meta[4 + 1]) { fprintf(stderr, "bundle totals\n"); abort(); }
        nrvh_free_bundle(&b);
      } else {
        rc = load_fast5_image(img, n, G, SG, (int)below(2), &r, err, sizeof err);
        if (rc == NRVH_OK) {
          if (r.n_ev < 2 || r.n_raw < 0 || !r.raw || !r.starts || !r.feat || !r.bases) { fprintf(stderr, "OK with a broken record\n"); abort(); }
          for (int64_t i = 0; i < r.n_ev; ++i) if (r.starts[i] < 0) { fprintf(stderr, "negative start in an accepted read\n"); abort(); }
          if (it % 64 < 8) finish_some(&r, tmpdir);
          nrvh_free_read(&r);
        }
      }
      free(img);
      if (rc < 0 || rc > 4) { fprintf(stderr, "return code %d\n", rc); abort(); }
      ++g_codes[rc]; ++per_class[cls]; ok_class[cls] += rc == NRVH_OK;
      ++total;
    }
    free(d); free(orig);
    printf("%s: %d mutations; accepted per class", strrchr(files[fi], '/') ? strrchr(files[fi], '/') + 1 : files[fi], nmut);
    for (int c = 0; c < 8; ++c) printf(" %ld/%ld", ok_class[c], per_class[c]);
    printf("\n");
  }
  printf("fuzzed %ld  ok %ld unsupported %ld e_read %ld e_io %ld e_arg %ld\n", total, g_codes[0], g_codes[1], g_codes[2], g_codes[3], g_codes[4]);
  return 0;
}

/* argument edge cases of the public entry points */
static int mode_api(const char* tmpdir, int nfiles, char** files) {
  char dst[600], err[NRVH_ERR_LEN];
  nrvh_read r;
  nrvh_bundle b;
  snprintf(dst, sizeof dst, "%s/api_out.fasta", tmpdir);
  if (nrvh_load_fast5(0, G, SG, 0, &r, err, sizeof err) != NRVH_E_ARG) abort();
  if (nrvh_load_fast5(files[0], G, SG, 0, 0, err, sizeof err) != NRVH_E_ARG) abort();
  if (nrvh_load_fast5("/nonexistent/x", G, SG, 0, &r, 0, 0) != NRVH_E_IO) abort();
  if (nrvh_load_fast5(files[0], "no_such_group", SG, 0, &r, err, 1) != NRVH_UNSUPPORTED) abort();
  if (nrvh_load_fast5(tmpdir, G, SG, 0, &r, err, sizeof err) == NRVH_OK) abort();           /* a directory */
  nrvh_free_read(0); nrvh_free_bundle(0);
  if (nrvh_load_bundle(0, 1, G, SG, 0, &b) != NRVH_E_ARG) abort();
  if (nrvh_load_bundle((const char* const*)files, 0, G, SG, 1, &b) != NRVH_OK || b.n_ok != 0 || b.n_ev != 0) abort();
  nrvh_free_bundle(&b);
  if (nrvh_load_bundle((const char* const*)files, nfiles, G, SG, 1, &b) != NRVH_OK || b.n_ok != nfiles) abort();
  /* the bundle through the bundle finisher, FASTA and FASTQ, T = 11 */
  {
    const int T = 11;
    int64_t* ev_len = (int64_t*)malloc((size_t)nfiles * 8);
    int64_t* nw = (int64_t*)malloc((size_t)nfiles * 8);
    int32_t* st = (int32_t*)malloc((size_t)nfiles * 4);
    const char** names = (const char**)malloc((size_t)nfiles * sizeof(char*));
    const char** dsts = (const char**)malloc((size_t)nfiles * sizeof(char*));
    char* dbuf = (char*)malloc((size_t)nfiles * 600);
    for (int i = 0; i < nfiles; ++i) {
      ev_len[i] = (int64_t)b.meta[4 * i + 1];
      names[i] = "n";
      snprintf(dbuf + 600 * i, 600, "%s/b%d_out.fastq", tmpdir, i);
      dsts[i] = dbuf + 600 * i;
    }
    int8_t* a1 = (int8_t*)calloc((size_t)b.n_ev + 1, 1);
    int8_t* a2 = (int8_t*)calloc((size_t)b.n_ev + 1, 1);
    uint8_t* qc = (uint8_t*)malloc((size_t)b.n_ev + 1);
    rng_s = 99;
    for (int64_t i = 0; i < b.n_ev; ++i) { a1[i] = (int8_t)below(6); a2[i] = (int8_t)below(5); qc[i] = (uint8_t)(33 + below(60)); }
    if (nrvh_finish_bundle(b.bases, ev_len, nfiles, a1, a2, b.n_ev, T, qc, names, dsts, 1, nw, st) != NRVH_OK) abort();
    for (int i = 0; i < nfiles; ++i) if (st[i] != NRVH_OK) abort();
    if (nrvh_finish_bundle(b.bases, ev_len, nfiles, a1, a2, b.n_ev, T, 0, names, dsts, 0, 0, st) != NRVH_OK) abort();
    /* fewer windows than the reads need: those reads are refused one by one, nothing is read past n_win_total */
    if (nrvh_finish_bundle(b.bases, ev_len, nfiles, a1, a2, ev_len[0] / 2, T, 0, names, dsts, 0, nw, st) != NRVH_OK) abort();
    if (st[0] != NRVH_E_ARG) abort();
    ev_len[0] = -5;
    if (nrvh_finish_bundle(b.bases, ev_len, nfiles, a1, a2, b.n_ev, T, 0, names, dsts, 0, nw, st) != NRVH_OK || st[0] != NRVH_E_ARG) abort();
    if (nrvh_finish_bundle(b.bases, ev_len, nfiles, a1, a2, b.n_ev, 0, 0, names, dsts, 0, nw, st) != NRVH_E_ARG) abort();
    /* out-of-range classes are clamped, not indexed */
    for (int64_t i = 0; i < 64 && i < b.n_ev; ++i) { a1[i] = (int8_t)(i % 2 ? 127 : -128); a2[i] = (int8_t)(i % 3 ? 126 : -100); }
    if (nrvh_finish_read(b.bases, 200, a1, a2, 189, T, 0, "x", dst, 0, nw) != NRVH_OK) abort();
    if (nrvh_finish_read(b.bases, 5, a1, a2, 0, T, 0, "short", dst, 0, nw) != NRVH_OK || nw[0] != 5) abort();
    if (nrvh_finish_read(b.bases, 0, 0, 0, 0, T, 0, "empty", dst, 1, nw) != NRVH_OK || nw[0] != 0) abort();
    if (nrvh_finish_read(b.bases, 10, a1, a2, 11, T, 0, "x", dst, 0, nw) != NRVH_E_ARG) abort();
    if (nrvh_finish_read(b.bases, 10, a1, a2, 1, T, 0, "x", "/nonexistent/dir/x", 0, nw) != NRVH_E_IO) abort();
    free(ev_len); free(nw); free(st); free(names); free(dsts); free(dbuf); free(a1); free(a2); free(qc);
  }
  nrvh_free_bundle(&b);
  /* nrvh_event_stats: empty, clipped and out-of-range events */
  {
    int16_t raw[40];
    for (int i = 0; i < 40; ++i) raw[i] = (int16_t)(i * 37 - 500);
    const int32_t starts[6] = {0, 3, 3, 20, 39, 400};
    double mean[6], sd[6];
    if (nrvh_event_stats(raw, 40, starts, 6, 5, mean, sd)) abort();
    if (!(mean[1] != mean[1]) || !(mean[5] != mean[5])) abort();       /* empty ranges: NaN */
    if (nrvh_event_stats(raw, 40, starts, 0, 5, 0, 0)) abort();
    if (nrvh_event_stats(0, 0, starts, 6, 5, mean, sd)) abort();
    if (nrvh_event_stats(raw, -1, starts, 6, 5, mean, sd) != -1) abort();
    const int32_t neg[2] = {-7, 2147483647};
    if (nrvh_event_stats(raw, 40, neg, 2, 2147483647, mean, sd)) abort();
  }
  printf("api ok\n");
  return 0;
}

/* ---- threads: what cli's pool does - loaders and finishers of one process at once ------------------------------------------- */
typedef struct { int id, iters, nfiles; char** files; const char* tmpdir; int64_t n_ev; int failed; } Job;
static void* thread_main(void* p) {
  Job* j = (Job*)p;
  char own[600], shared[600];
  snprintf(own, sizeof own, "%s/t%d_out.fasta", j->tmpdir, j->id);
  snprintf(shared, sizeof shared, "%s/shared_out.fasta", j->tmpdir);
  for (int it = 0; it < j->iters; ++it) {
    nrvh_bundle b;
    if (nrvh_load_bundle((const char* const*)j->files, j->nfiles, G, SG, it & 1, &b) != NRVH_OK || b.n_ok != j->nfiles) { j->failed = 1; return 0; }
    j->n_ev = b.n_ev;
    const int64_t el = (int64_t)b.meta[1], nwin = el - 11;
    int8_t* a = (int8_t*)calloc((size_t)el, 1);
    for (int64_t i = 0; i < nwin; ++i) a[i] = (int8_t)(2 + (i + j->id) % 4);
    int64_t nw;
    if (nrvh_finish_read(b.bases, el, a, a, nwin, 11, 0, "t", it % 3 ? own : shared, 0, &nw) != NRVH_OK) j->failed = 1;
    free(a);
    nrvh_free_bundle(&b);
  }
  return 0;
}
static int mode_threads(int nthreads, int iters, const char* tmpdir, int nfiles, char** files) {
  pthread_t th[64];
  Job jobs[64];
  if (nthreads > 64) nthreads = 64;
  for (int i = 0; i < nthreads; ++i) {
    jobs[i] = (Job){i, iters, nfiles, files, tmpdir, 0, 0};
    if (pthread_create(&th[i], 0, thread_main, &jobs[i])) return 2;
  }
  for (int i = 0; i < nthreads; ++i) pthread_join(th[i], 0);
  for (int i = 0; i < nthreads; ++i) if (jobs[i].failed || jobs[i].n_ev != jobs[0].n_ev) { fprintf(stderr, "thread %d disagrees\n", i); return 1; }
  printf("threads ok: %d x %d bundles of %d files, %lld bases each\n", nthreads, iters, nfiles, (long long)jobs[0].n_ev);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 6 && !strcmp(argv[1], "fuzz")) return mode_fuzz(atoi(argv[2]), strtoull(argv[3], 0, 10), argv[4], argc - 5, argv + 5);
  if (argc >= 4 && !strcmp(argv[1], "api")) return mode_api(argv[2], argc - 3, argv + 3);
  if (argc >= 6 && !strcmp(argv[1], "threads")) return mode_threads(atoi(argv[2]), atoi(argv[3]), argv[4], argc - 5, argv + 5);
  fprintf(stderr, "usage: host_fuzz fuzz N SEED TMPDIR file... | api TMPDIR file... | threads NTHREADS ITERS TMPDIR file...\n");
  return 2;
}
