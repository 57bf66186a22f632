"""Synthetic vg-like k-mer batches for benchmarks and size-independent parity tests.

Follows SURVEY.md section 8(d): region i = chr22:(16_000_000 + 1000 i)-(... + 200); per region a
200-bp reference haplotype drawn i.i.d. from bg_nt frequencies; the 200-W+1 forward windows and
their reverse complements ('-' rows, START > STOP) are the ``ref`` rows; the remaining rows are
``non.ref`` variants of random reference windows with 1-2 substitutions; 1 % of the rows are
overwritten by a sample from the motif's own PWM columns; 0.1 % get one ``N``;
haplotype_frequency is uniform in [0, 5096] with 5 % forced to 0.
RNG: numpy.random.Generator(PCG64(seed)), seed = 20240139 + config index (+ rank offset).
"""
from dataclasses import dataclass

import numpy as np

BG_NT = np.array([0.2951, 0.2047, 0.2048, 0.2955])  # A C G T (tutorials/.../bg_nt)
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b

REGION_LEN = 200
REGION_STRIDE = 1000
REGION_ORIGIN = 16_000_000


@dataclass
class KmerBatch:
    kmers: np.ndarray        # uint8 [n, W] ASCII
    region: np.ndarray       # int32 [n]   region index (global)
    start: np.ndarray        # int64 [n]
    stop: np.ndarray         # int64 [n]
    strand: np.ndarray       # uint8 [n]   '+' / '-'
    freq: np.ndarray         # int64 [n]
    is_ref: np.ndarray       # uint8 [n]
    width: int

    def __len__(self):
        return int(self.kmers.shape[0])

    def region_name(self, i):
        s = REGION_ORIGIN + REGION_STRIDE * int(i)
        return f"chr22:{s}-{s + REGION_LEN}"


def seed_for(config_index: int, rank: int = 0) -> int:
    return 20240139 + int(config_index) + 1_000_003 * int(rank)


def make_batch(n_regions: int, rows_per_region: int, width: int, pwm_probs: np.ndarray,
               seed: int, region_base: int = 0, block: int = 512) -> KmerBatch:
    """pwm_probs: f64 [4, W] column-stochastic (rows A,C,G,T) used for the 1 % planted rows."""
    W = int(width)
    nwin = REGION_LEN - W + 1
    n_ref = 2 * nwin
    if rows_per_region < n_ref:
        raise ValueError("rows_per_region must hold the forward and reverse reference windows")
    n_var = rows_per_region - n_ref
    rng = np.random.Generator(np.random.PCG64(seed))
    cdf = np.cumsum(pwm_probs / pwm_probs.sum(0, keepdims=True), axis=0)  # [4, W]
    out_k, out_reg, out_st, out_sp, out_sd, out_fr, out_rf = [], [], [], [], [], [], []
    win_idx = np.arange(nwin)[:, None] + np.arange(W)[None, :]            # [nwin, W]
    for r0 in range(0, n_regions, block):
        R = min(block, n_regions - r0)
        hap = _ACGT[rng.choice(4, size=(R, REGION_LEN), p=BG_NT / BG_NT.sum())]          # [R, 200]
        fwd = hap[:, win_idx]                                             # [R, nwin, W]
        rev = _COMP[fwd[:, :, ::-1]]
        origin = REGION_ORIGIN + REGION_STRIDE * (region_base + r0 + np.arange(R, dtype=np.int64))
        f_start = origin[:, None] + np.arange(nwin, dtype=np.int64)[None, :]
        f_stop = f_start + W
        # variants: a random reference window on a random strand with 1-2 substitutions
        src = rng.integers(0, nwin, size=(R, n_var))
        neg = rng.random((R, n_var)) < 0.5
        var = np.take_along_axis(fwd, src[:, :, None], axis=1)            # [R, n_var, W]
        var = np.where(neg[:, :, None], _COMP[var[:, :, ::-1]], var)
        nsub = rng.integers(1, 3, size=(R, n_var))
        for k in range(2):
            pos = rng.integers(0, W, size=(R, n_var))
            base = _ACGT[rng.integers(0, 4, size=(R, n_var))]
            apply = nsub > k
            cur = np.take_along_axis(var, pos[:, :, None], axis=2)[:, :, 0]
            np.put_along_axis(var, pos[:, :, None], np.where(apply, base, cur)[:, :, None], axis=2)
        v_start = origin[:, None] + src
        v_stop = v_start + W
        km = np.concatenate([fwd, rev, var], axis=1).reshape(R * rows_per_region, W)
        st = np.concatenate([f_start, f_stop, np.where(neg, v_stop, v_start)], axis=1).reshape(-1)
        sp = np.concatenate([f_stop, f_start, np.where(neg, v_start, v_stop)], axis=1).reshape(-1)
        sd = np.concatenate([np.full((R, nwin), ord("+"), np.uint8), np.full((R, nwin), ord("-"), np.uint8),
                             np.where(neg, ord("-"), ord("+")).astype(np.uint8)], axis=1).reshape(-1)
        rf = np.concatenate([np.ones((R, n_ref), np.uint8), np.zeros((R, n_var), np.uint8)], axis=1).reshape(-1)
        n = km.shape[0]
        # 1 % planted motif instances
        plant = np.nonzero(rng.random(n) < 0.01)[0]
        if len(plant):
            u = rng.random((len(plant), W))
            code = (u[:, None, :] > cdf[None, :, :]).sum(1).clip(0, 3)    # inverse-CDF per column
            km[plant] = _ACGT[code]
            rf[plant] = 0
        # 0.1 % rows with one N
        nrow = np.nonzero(rng.random(n) < 0.001)[0]
        if len(nrow):
            km[nrow, rng.integers(0, W, size=len(nrow))] = ord("N")
        fr = rng.integers(0, 5097, size=n, dtype=np.int64)
        fr[rng.random(n) < 0.05] = 0
        out_k.append(km)
        out_reg.append(np.repeat(np.arange(region_base + r0, region_base + r0 + R, dtype=np.int32),
                                 rows_per_region))
        out_st.append(st); out_sp.append(sp); out_sd.append(sd); out_fr.append(fr); out_rf.append(rf)
    return KmerBatch(np.ascontiguousarray(np.concatenate(out_k)), np.concatenate(out_reg),
                     np.concatenate(out_st), np.concatenate(out_sp), np.concatenate(out_sd),
                     np.concatenate(out_fr), np.concatenate(out_rf), W)


def tsv_lines(batch: KmerBatch, order: np.ndarray) -> np.ndarray:
    """The rows `order` of the batch as vg-style TSV lines (numpy bytes array, NUL padded):
    REGION KMER CHR:START(+|-) CHR:STOP(+|-) COUNT ref|non.ref PATH.  Assembled with numpy's string
    ufuncs instead of a Python loop per row."""
    S = np.strings
    reg = batch.region[order]
    origin = REGION_ORIGIN + REGION_STRIDE * reg.astype(np.int64)
    name = S.add(S.add(S.add(b"chr22:", origin.astype("S")), b"-"), (origin + REGION_LEN).astype("S"))
    km = np.ascontiguousarray(batch.kmers[order]).view(f"S{batch.width}").ravel()
    sd = np.ascontiguousarray(batch.strand[order]).view("S1").ravel()
    tab = np.array(b"\t", dtype="S1")
    pos = lambda v: S.add(S.add(b"chr22:", v.astype("S")), sd)
    line = name
    for col in (km, pos(batch.start[order]), pos(batch.stop[order]), batch.freq[order].astype("S"),
                np.where(batch.is_ref[order] != 0, b"ref", b"non.ref").astype("S7"), S.add(S.add(b"1", sd), b",\n")):
        line = S.add(S.add(line, tab), col)
    return line


def tsv_text(batch: KmerBatch, rows=None) -> bytes:
    order = np.arange(len(batch)) if rows is None else np.asarray(rows)
    return tsv_lines(batch, order).tobytes().replace(b"\0", b"")


def write_tsv_dir(batch: KmerBatch, out_dir: str, regions_per_file: int = 1):
    """Write the batch as vg-style TSVs under out_dir/width_W/, `regions_per_file` regions to a file (vg
    writes one file per region; the ingest takes any grouping)."""
    import os
    d = os.path.join(out_dir, f"width_{batch.width}")
    os.makedirs(d, exist_ok=True)
    order = np.argsort(batch.region, kind="stable")
    reg = batch.region[order]
    line = tsv_lines(batch, order)
    bounds = np.concatenate([[0], np.nonzero(np.diff(reg))[0] + 1, [len(reg)]])
    step = max(1, int(regions_per_file))
    for k in range(0, len(bounds) - 1, step):
        lo, hi = bounds[k], bounds[min(k + step, len(bounds) - 1)]
        fname = batch.region_name(int(reg[lo])).replace(":", "_") + ".tsv"
        with open(os.path.join(d, fname), "wb") as fh:
            fh.write(line[lo:hi].tobytes().replace(b"\0", b""))
    return d


# ---------------------------------------------------------------------------- device-side generators
def make_device_kmers(n: int, width: int, pwm_probs: np.ndarray, seed: int, device):
    """uint8 [n, W] k-mers generated ON the device for the configs that do not fit a host array
    (BASELINE config 3's per-GPU shard, configs 4 and 5): i.i.d. bg_nt bases, 1 % of the rows a sample
    of the motif's own PWM columns, 0.1 % of the rows with one N.  torch is used as an RNG + buffer only."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    alpha = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    bg = torch.tensor(BG_NT / BG_NT.sum(), device=device)
    W = int(width)
    out = torch.empty((n, W), dtype=torch.uint8, device=device)
    step = 25_000_000
    for a in range(0, n, step):
        b = min(n, a + step)
        idx = torch.multinomial(bg, (b - a) * W, replacement=True, generator=g).view(b - a, W)
        out[a:b] = alpha[idx]
    k = max(1, n // 100)
    rows = torch.randint(0, n, (k,), generator=g, device=device)
    cdf = torch.tensor(np.cumsum(pwm_probs / pwm_probs.sum(0, keepdims=True), axis=0), device=device)   # [4, W]
    for a in range(0, k, 1_000_000):
        u = torch.rand((min(k, a + 1_000_000) - a, W), generator=g, device=device, dtype=torch.float64)
        code = (u[:, None, :] > cdf[None, :, :]).sum(1).clamp(max=3)
        out[rows[a:a + 1_000_000]] = alpha[code]
    if n >= 1000:
        out[torch.randint(0, n, (n // 1000,), generator=g, device=device), W // 2] = ord("N")
    return out


def jaspar_style_probs(width: int, rng: np.random.Generator, pseudo: float = 0.1, bg=None) -> np.ndarray:
    """Synthetic JASPAR-style motif: columns ~ Dirichlet(0.3) * 1000 rounded
    to counts, turned into probabilities with the count-format pseudocount rule (pyx:192-261)."""
    counts = np.rint(rng.dirichlet([0.3] * 4, size=int(width)).T * 1000)
    tot = counts.sum(0)
    bg = np.full(4, 0.25) if bg is None else np.asarray(bg, dtype=np.float64)
    return (counts / tot * tot.astype(int) + pseudo * bg[:, None]) / (tot.astype(int) + pseudo)


def synthetic_motif(width: int, rng: np.random.Generator, bg) -> dict:
    """JASPAR-style synthetic PWM through the library's own log-odds + scaling (host side of
    libgrafimo_hip.so, no GPU): -> dict(sm, bg, min_val, scale, offset, probs, width)."""
    from .device import compute_log_odds_dense, scale_pwm_dense
    bg = (np.asarray(bg, dtype=np.float64) + 5e-7) / (np.sum(bg) + 2e-6)     # norm_bg (motif_ops.py:1268-1302)
    probs = jaspar_style_probs(width, rng, 0.1, bg)
    sm, mn, mx, scale, offset = scale_pwm_dense(compute_log_odds_dense(probs, bg))
    return dict(sm=sm, bg=bg, min_val=mn, scale=scale, offset=float(offset), probs=probs, width=int(width))


class SyntheticMotif:
    """A synthetic_motif() record with the members of the reference's Motif that the scoring entry points read
    (grafimo_amd.motif.MOTIF_FIELDS): what bench.py hands to compute_results_from_graph[_many].  No pval_matrix: the score
    distribution is computed on the device."""

    def __init__(self, rec: dict, name: str):
        self.score_matrix = rec["sm"]
        self.nucsmap = {n: i for i, n in enumerate("ACGT")}
        self.bg = {n: float(rec["bg"][i]) for i, n in enumerate("ACGT")}
        self.min_val, self.scale, self.offset = int(rec["min_val"]), int(rec["scale"]), np.double(rec["offset"])
        self.width = int(rec["width"])
        self.motif_id, self.motif_name = name, name.lower()


def motif_object(rec: dict, name: str) -> SyntheticMotif:
    return SyntheticMotif(rec, name)


def config_motifs(cfg: int):
    """The synthetic motifs of BASELINE config 4 (one W=30 PWM, uniform background) and config 5 (fifty PWMs,
    widths cycling 8..25, per-motif background ~ Dirichlet(50 bg_nt)), seeded as SURVEY 8(d) says
    (PCG64(20240139 + config index)); bench.py and the full-size parity tests share them."""
    rng = np.random.default_rng(20240139 + int(cfg))
    if cfg == 4:
        return [synthetic_motif(30, rng, np.full(4, 0.25))]
    if cfg == 5:
        out = []
        for k in range(50):
            bg = rng.dirichlet(50 * BG_NT)
            out.append(synthetic_motif(8 + (k % 18), rng, bg))
        return out
    raise ValueError("config_motifs: config 4 or 5")


def make_graph_index(n_regions: int, width: int = 19, n_haplotypes: int = 5096, site_every: int = 32,
                     del_frac: float = 0.06, seed: int = 20240139, with_counts: bool = True, with_dels: bool = True,
                     plant=None):
    """A synthetic chromosome for the extraction kernels at BASELINE config-2 scale: region i = [16 000 + 1000 i,
    + 200] (SURVEY 8d's layout, shifted to a short contig), variant sites at 1000-Genomes-like density (one per
    `site_every` bp, 3 % with two alternates, `del_frac` of them deletions of 1..8 bases kept apart from each other),
    allele frequencies skewed to rare variants (af = u^4), one haplotype bitset per alternate allele.  Sites are
    drawn only where a window of a region can see them.  `plant` = (probs f64 [4, W], fraction): in that fraction of the
    regions one window of the reference is overwritten by a sample from the PWM's columns -- SURVEY 8(d)'s "1 % of the rows are
    overwritten by a sample from the motif's own PWM", so that a strict threshold (q < 1e-4) still reports rows; drawn from a
    generator of its own: everything else of the graph is what it is without.  -> (GraphIndex, [(S, E)] regions)."""
    from .extract_regions import GraphIndex
    rng = np.random.default_rng(seed)
    L = 16_000 + REGION_STRIDE * n_regions + 4 * REGION_LEN
    ref = _ACGT[rng.choice(4, size=L, p=BG_NT / BG_NT.sum())]
    regions = [(16_000 + REGION_STRIDE * i, 16_000 + REGION_STRIDE * i + REGION_LEN) for i in range(n_regions)]
    if plant is not None:
        probs, frac = plant
        probs = np.asarray(probs, dtype=np.float64)
        Wp = probs.shape[1]
        rng_p = np.random.default_rng(seed + 7919)
        chosen = np.flatnonzero(rng_p.random(n_regions) < frac)
        cdf = np.cumsum(probs / probs.sum(0, keepdims=True), axis=0)
        u = rng_p.random((len(chosen), Wp))
        sample = _ACGT[(u[:, None, :] > cdf[None, :, :]).sum(1).clip(0, 3)]            # [n, Wp]
        off = rng_p.integers(0, REGION_LEN - Wp + 1, size=len(chosen))
        for k, r in enumerate(chosen.tolist()):
            ref[regions[r][0] + int(off[k]):regions[r][0] + int(off[k]) + Wp] = sample[k]
    span = REGION_LEN + 2 * width + 16
    per_region = max(1, span // site_every)
    starts = np.asarray([r[0] for r in regions], dtype=np.int64) - width - 8
    pos = (starts[:, None] + np.sort(rng.integers(0, span, size=(n_regions, per_region)), axis=1)).ravel()
    pos = np.unique(pos).astype(np.int32)
    V = len(pos)
    n_alts = np.where(rng.random(V) < 0.97, 1, 2).astype(np.uint8)
    alt_bases = np.zeros((V, 3), np.uint8)
    code = np.searchsorted(_ACGT, ref[pos])
    for a in range(2):
        alt_bases[:, a] = _ACGT[(code + 1 + a) % 4]
    alt_bases[n_alts < 2, 1] = 0
    del_len = np.zeros(V, np.int32)
    if with_dels:
        last_end = -1
        for i in np.nonzero(rng.random(V) < del_frac)[0]:
            ln = int(rng.integers(1, 9))
            if pos[i] > last_end and pos[i] + ln < L - 1:
                del_len[i] = ln
                last_end = int(pos[i]) + ln
        n_alts[del_len > 0] = 1
        alt_bases[del_len > 0] = 0
    bits = None
    if with_counts:
        hw = (n_haplotypes + 63) // 64
        bits = np.zeros((V, 3, hw), np.uint64)
        af = rng.random(V) ** 4
        for s in range(0, V, 8192):
            e = min(V, s + 8192)
            for a in range(2):
                carry = rng.random((e - s, hw * 64)) < (af[s:e, None] if a == 0 else 0.5 * af[s:e, None])
                carry[:, n_haplotypes:] = False
                if a == 1:      # a haplotype carries one alternate at most
                    carry &= ~np.unpackbits(bits[s:e, 0, :].view(np.uint8), axis=1, bitorder="little").astype(bool)
                    carry[n_alts[s:e] < 2] = False
                bits[s:e, a, :] = np.packbits(carry, axis=1, bitorder="little").view(np.uint64)
    idx = GraphIndex("22", ref, pos, n_alts, alt_bases, bits, n_haplotypes if with_counts else 0,
                     del_len=del_len if with_dels else None)
    return idx, regions
