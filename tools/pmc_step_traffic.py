"""Reduce the rocprofv3 --pmc passes of tools/pmc_step_traffic.sh to per-launch L2<->fabric traffic of the GEMM kernels."""
import glob, json, os, sqlite3, sys, collections

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


COLS = []
PER = collections.defaultdict(lambda: collections.defaultdict(float))  # kernel -> counter -> sum


def load(d):
    """-> ({counter: sum over the GEMM dispatches}, {counter: number of DISTINCT dispatches}).  (A dispatch has one row per counter
    and hardware instance - XCC / channel - so rows are summed and dispatches are counted by their id.)"""
    out = collections.defaultdict(float)
    seen = collections.defaultdict(set)
    per = PER
    for db in glob.glob(d + "/**/*.db", recursive=True):
        c = sqlite3.connect(db)
        cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
        COLS[:] = cols
        ix = {k: i for i, k in enumerate(cols)}
        name_col = "kernel_name" if "kernel_name" in ix else "name"
        id_col = next((k for k in ("dispatch_id", "kernel_dispatch_id", "correlation_id", "stack_id", "id") if k in ix), None)
        for j, r in enumerate(c.execute("select * from counters_collection")):
            if "gemm_nt" not in str(r[ix[name_col]]) and "gemm_w4" not in str(r[ix[name_col]]):
                continue
            out[r[ix["counter_name"]]] += float(r[ix["value"]])
            seen[r[ix["counter_name"]]].add(r[ix[id_col]] if id_col else j)
            kn = str(r[ix[name_col]]).replace("mhgemm::", "").replace("(anonymous namespace)::", "").split("(")[0]
            per[kn][r[ix["counter_name"]]] += float(r[ix["value"]])
    return out, {k: len(v) for k, v in seen.items()}


a, na = load(sys.argv[1]); b, nb = load(sys.argv[2]); c, nc = load(sys.argv[3]) if len(sys.argv) > 3 else ({}, {})
launches = max(na.values()) if na else 0
rd, rd32 = a.get("TCC_EA0_RDREQ_sum", 0.0), a.get("TCC_EA0_RDREQ_32B_sum", 0.0)
if c.get("TCC_EA0_RDREQ_128B_sum") or c.get("TCC_EA0_RDREQ_64B_sum"):
    r128, r64 = c.get("TCC_EA0_RDREQ_128B_sum", 0.0), c.get("TCC_EA0_RDREQ_64B_sum", 0.0)
    read_bytes = 32 * rd32 + 64 * r64 + 128 * r128
    how = "reads = 32*RDREQ_32B + 64*RDREQ_64B + 128*RDREQ_128B (exact request sizes)"
else:
    read_bytes = 32 * rd32 + 128 * (rd - rd32)
    how = "reads = 32*RDREQ_32B + 128*(RDREQ - RDREQ_32B): wide coalesced reads are 128-B requests on gfx950 (the guide's 2x FETCH_SIZE correction)"
wr, wr64 = b.get("TCC_EA0_WRREQ_sum", 0.0), b.get("TCC_EA0_WRREQ_64B_sum", 0.0)
write_bytes = 64 * wr64 + 32 * (wr - wr64)
hit, miss = a.get("TCC_HIT_sum", 0.0), a.get("TCC_MISS_sum", 0.0)
L = max(1, launches)
import bench  # noqa: E402  (kernel_source_stamp, algorithmic byte model)


CFG5 = os.environ.get("PMC_CONFIG", "cfg3") == "cfg5"  # cfg 5: the same 32 768 tokens per step (B 4 x S 8192), 16 images, decoder + head operands are 1-byte e4m3


def algorithmic_bytes_per_step():
    """A + B + C moved once per GEMM of one cfg-3 training step (SURVEY §8d shapes): what bench.py's `algorithmic_gb_per_launch` sums.
    PMC_CONFIG=cfg5: the fp8 step - decoder / lm_head operands are one byte per element (the quantised copies), outputs as in the bf16 step."""
    T, d, ff, V, L = 8 * 4096, 4096, 11008, 32064, 32
    osz = 1.0 if CFG5 else 2.0
    def g(M, N, K, csize=2):
        return osz * (M * K + N * K) + csize * M * N
    fwd = g(T, 3 * d, d) + g(T, d, d) + 2.0 * (T * d + 2 * ff * d + 3 * T * ff) + g(T, d, ff)
    dgrad = g(T, ff, d) + 2.0 * 3 * T * ff + g(T, d, 2 * ff) + g(T, d, d) + g(T, d, 3 * d)
    wgrad = g(d, ff, T) + g(2 * ff, d, T) + g(d, d, T) + g(3 * d, d, T)
    head = g(T, V, d, 4) + g(T, d, V) + g(V, d, T)
    Tv, vd, vff = (16 if CFG5 else 48) * 577, 1024, 4096
    osz = 2.0  # (the CLIP tower stays 16-bit in the fp8 step)
    vit = 23 * (g(Tv, 3 * vd, vd) + g(Tv, vd, vd) + g(Tv, vff, vd) + g(Tv, vd, vff)) * 3 + g(Tv, vd, 640) * 2 + g(Tv, d, vd) * 3
    return L * (fwd + dgrad + wgrad) + head + vit


def _kernel_rows():
    rows = {}
    for kn, c in PER.items():
        rd = 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0.0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0.0) + 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0.0)
        wrq, wr64q = c.get("TCC_EA0_WRREQ_sum", 0.0), c.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        h, m = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
        rows[kn] = {"fabric_read_gb_per_step": round(rd / 2 / 1e9, 1), "fabric_write_gb_per_step": round((64 * wr64q + 32 * (wrq - wr64q)) / 2 / 1e9, 1),
                    "l2_hit_rate": round(h / max(1.0, h + m), 4)}
    return rows


print(json.dumps({"per_kernel": _kernel_rows(), "kernel": ("scaled-fp8 + bf16 MFMA GEMM kernels (all kernel launches of the two cfg-5 training steps of `bench.py --config cfg5 --steps 1 --warmup 1`)" if CFG5 else
                             "bf16 MFMA GEMM kernels (all kernel launches of the two cfg-3 training steps of `bench.py --steps 1 --warmup 1`)"), "launches": launches,
                  "steps_profiled": 2, "traffic_bytes_per_step": (read_bytes + write_bytes) / 2, "algorithmic_bytes_per_step": algorithmic_bytes_per_step(),
                  "traffic_over_algorithmic": (read_bytes + write_bytes) / 2 / algorithmic_bytes_per_step(),
                  "kernel_source_stamp": bench.kernel_source_stamp(), "algorithmic_bytes_per_launch": 2 * algorithmic_bytes_per_step() / L,
                  "fabric_read_bytes_per_launch": read_bytes / L, "fabric_write_bytes_per_launch": write_bytes / L,
                  "traffic_bytes_per_launch": (read_bytes + write_bytes) / L, "l2_hit_rate": hit / max(1.0, hit + miss),
                  "method": "rocprofv3 --pmc, separate passes (tools/pmc_step_traffic.sh); " + how + "; writes = 64*WRREQ_64B + 32*(WRREQ-WRREQ_64B); "
                            "counters sit on the L2<->fabric (EA) side, i.e. Infinity-Cache hits are included"}, indent=1))
