"""Level shapes and row-slab partition arithmetic (host only, no GPU needed).

Two layouts appear in this package:

* the reference's: four square sub-domain blocks of side 2**(n-i)+2 per level i < n_para plus
  whole-domain levels of side 2**(n-i)+1 for i = n_para-1 .. n-1
  (Multigrid.__init__, core/parallel_multigrid.py:182-219 of the reference checkout);
* this package's: the global (NY x NX) grid, boundary included, cut into P row slabs for the
  finest Jacobi levels; the coarser Jacobi levels and every whole-domain level are held whole by
  every rank.

`reference_layout` restates the first (used by the drop-in `Multigrid` and by tests),
`SlabPlan` describes the second.
"""
from dataclasses import dataclass, field
from typing import List


def reference_layout(n, n_para, h0=1.0):
    """Shapes and mesh widths of every level the reference allocates for (n, n_para)."""
    if not n_para < n:
        raise AssertionError("n_para < n")  # parallel_multigrid.py:159
    wl = [2 ** (n - i) + 1 for i in range(n_para - 1, n)]
    h_wl = [h0 * 2 ** (i + 1) for i in range(n_para - 1, n)]
    sb = [2 ** (n - i) + 2 for i in range(n_para)]
    h_sb = [h0 * 2 ** i for i in range(n_para)]
    b_sb = sb + [2 ** (n - n_para) + 2]
    return {
        "x_sb": [[m, m] for m in sb], "r_sb": [[m, m] for m in sb], "b_sb": [[m, m] for m in b_sb],
        "x_wl": [[m, m] for m in wl], "b_wl": [[m, m] for m in wl], "h_sb": h_sb, "h_wl": h_wl,
    }


# offsets of the four block ranks (parallel_multigrid.py:222-228) and their neighbours (:49-56)
BLOCK_OFFSETS = [{"j": 0, "i": 0}, {"j": 0, "i": -1}, {"j": -1, "i": 0}, {"j": -1, "i": -1}]
BLOCK_NEIGHBOURS = [{"east": 1, "north": 2}, {"west": 0, "north": 3}, {"east": 3, "south": 0},
                    {"west": 2, "south": 1}]


def pitch(nx, align=16):
    """Leading dimension for nx columns: a multiple of 16 doubles (128 B) so that every row starts
    on a cache-line boundary and the widest vector loads are aligned."""
    return (nx + align - 1) // align * align


@dataclass
class SlabLevel:
    """One Jacobi level on one rank.  The local array has `halo` rows above and below the `rows` owned
    ones: local row t holds global row rank*rows - halo + t.  On rank 0 the first owned row is the
    global Dirichlet row (never updated; the rows above it are unused); the last rank additionally
    holds the bottom Dirichlet row as local row halo + rows."""
    level: int
    rows: int          # owned rows R_l
    halo: int          # halo depth H (rows per side)
    ny: int            # local rows = rows + 2 * halo
    nx: int            # global columns, boundary included
    ld: int
    h: float
    t0: int            # first local row that is updated
    t1: int            # one past the last updated local row
    fix_lo: int        # local row of the top Dirichlet row (rows <= it never change), -1 if none
    fix_hi: int        # local row of the bottom Dirichlet row, ny if none
    own0: int          # first owned local row, Dirichlet rows included
    own1: int          # one past the last owned local row
    gny: int           # global rows at this level

    @property
    def oj(self):
        """Row offset of the transfer operators: local fine row of local coarse row T is 2T + oj."""
        return -self.halo


@dataclass
class WholeLevel:
    """A grid every rank holds whole (the reference's x_wl/b_wl/r_wl levels)."""
    ny: int
    nx: int
    ld: int
    h: float


@dataclass
class SlabPlan:
    gny: int
    gnx: int
    n_para: int
    world: int
    rank: int
    h0: float
    halo: int = 1
    levels: List[SlabLevel] = field(default_factory=list)  # 0 .. n_para (the last one is the staging level)
    whole: List[WholeLevel] = field(default_factory=list)

    @property
    def unknowns(self):
        return (self.gny - 2) * (self.gnx - 2)


def make_plan(gny, gnx, n_para, world=1, rank=0, h0=1.0, halo=None, n_dist=None):
    """Partition a (gny x gnx) grid (boundary included) for `n_para` Jacobi levels over `world` ranks.

    Requirements: gny-1 = world * R with R divisible by 2**n_para (so that coarse row g/2 lives on
    the rank that owns fine row g on every Jacobi level, and slab boundaries fall on even rows),
    gnx-1 divisible by 2**n_para, and at least 2 cells per side on the first whole-domain level.
    `halo`: rows of halo per side; default 1 on one rank (nothing is exchanged), 8 on several (deep
    enough for a fused pass of 6 sweeps + residual, and even, which keeps the fine/coarse row
    correspondence t = 2T - halo on every level).
    `n_dist`: how many of the Jacobi levels (the finest ones) are actually cut into slabs; the
    per-rank requirements apply to those only (default: all).
    """
    n_dist = n_para if n_dist is None else n_dist
    if n_para < 1:
        raise ValueError("n_para >= 1 required")
    if (gny - 1) % world:
        raise ValueError(f"{gny - 1} rows of cells cannot be split evenly over {world} ranks")
    R0 = (gny - 1) // world
    if (gny - 1) % (1 << n_para) or (gnx - 1) % (1 << n_para):
        raise ValueError(f"rows-1 ({gny - 1}) and columns-1 ({gnx - 1}) must be divisible by 2**n_para = {1 << n_para}")
    if R0 % (1 << n_dist):
        raise ValueError(f"rows per rank ({R0}) must be divisible by 2**{n_dist} (slab boundaries on even rows of "
                         "every distributed level and of the first replicated one)")
    if ((gny - 1) >> n_para) < 2 or ((gnx - 1) >> n_para) < 2:
        raise ValueError("the first whole-domain level needs at least 2 cells per side")
    H = (1 if world == 1 else 8) if halo is None else halo
    if H != 1 and H % 2:
        raise ValueError("halo must be 1 or even")
    if world > 1 and (R0 >> (n_dist - 1)) < H:
        raise ValueError(f"the coarsest distributed level has {R0 >> (n_dist - 1)} rows per rank, fewer than the halo depth "
                         f"{H}: distribute fewer levels, use fewer ranks or a smaller halo")
    plan = SlabPlan(gny, gnx, n_para, world, rank, h0, H)
    last = rank == world - 1
    for l in range(n_para + 1):
        R = R0 >> l
        nx = ((gnx - 1) >> l) + 1
        ny = R + 2 * H
        plan.levels.append(SlabLevel(
            level=l, rows=R, halo=H, ny=ny, nx=nx, ld=pitch(nx), h=h0 * 2 ** l,
            t0=H + 1 if rank == 0 else H, t1=H + R,
            fix_lo=H if rank == 0 else -1, fix_hi=H + R if last else ny,
            own0=H, own1=H + R + (1 if last else 0), gny=((gny - 1) >> l) + 1))
    l = n_para
    while True:
        my, mx = (gny - 1) >> l, (gnx - 1) >> l
        if my < 2 or mx < 2 or (my << l) != gny - 1 or (mx << l) != gnx - 1:
            break
        plan.whole.append(WholeLevel(ny=my + 1, nx=mx + 1, ld=pitch(mx + 1), h=h0 * 2 ** l))
        l += 1
    return plan


def exchanges_per_cycle(n_dist, fused=True):
    """Communication rounds one V-cycle performs per rank with n_dist distributed Jacobi levels (see
    engine.SlabMultigrid._cycle), the max-reduction of the norm not counted.  Fused passes: per level
    one halo exchange of (x, correction) before the ascent pass and one of the restricted right-hand
    side after the descent pass (for the last distributed level that one is the all-gather; a level in
    the recompute scheme whose correction comes from a replicated level needs no exchange before its
    ascent pass, see SlabMultigrid.rounds_per_cycle for the exact count).  One kernel per operation:
    an exchange per sweep and one of r per restriction."""
    return 2 * n_dist if fused else 8 + 11 * (n_dist - 1) + 1


def algorithmic_bytes_per_cycle(plan, keep_ascent_residuals=False, fused_zero_sweep=True):
    """Compulsory HBM traffic of one V-cycle over the whole grid, in bytes (float64):
    Jacobi sweep 24 B/pt, residual 24, restriction 10 per fine point, prolongation 18 per fine
    point; the zero-fill of a coarse x is 0 when fused into the first sweep (which then moves 16
    B/pt instead of 24), else 8.  Whole-domain levels are included as their smoother traffic."""
    total = 0
    for l in range(plan.n_para):
        pts = (((plan.gny - 1) >> l) + 1) * (((plan.gnx - 1) >> l) + 1)
        if l == 0:
            per = 10 + 18 + 6 * 24 + 24
        else:
            first = 16 if fused_zero_sweep else 8 + 24
            per = first + 2 * 24 + 24 + 10 + 18 + 6 * 24 + (24 if keep_ascent_residuals else 0)
        total += per * pts
    return total


def own_bytes_per_cycle(plan, recomputed=()):
    """What the fused passes of one V-cycle themselves must move, in bytes (float64), every array element read or
    written once per pass: the finest pass 28 B/pt (reads x, b and a quarter of a point of coarse x, writes x and
    a quarter of coarse b); a coarse level with recomputed pre-smoothing 10 (descent: reads b, writes coarse b)
    + 18 (ascent: reads b and coarse x, writes x); a coarse level that stores its pre-smoothed x 18 + 26.
    `recomputed`: the levels that run in the recompute scheme.  The whole-domain levels (a few KB) are left out."""
    total = 0
    for l in range(plan.n_para):
        pts = (((plan.gny - 1) >> l) + 1) * (((plan.gnx - 1) >> l) + 1)
        total += (28 if (l == 0 or l in recomputed) else 44) * pts
    return total


SOR_LDS_DOUBLES = 20000  # MG_SOR_LDS_DOUBLES of include/mg_hip.h


def whole_level_fits_lds(ny, nx):
    """True when the lexicographic SSOR of a (ny x nx) whole-domain level runs LDS-resident (the
    kernels that take ~0.1-0.2 us per wavefront step); beyond that it walks global memory, an
    order of magnitude slower per step."""
    return ny * ((nx + 1) & ~1) <= SOR_LDS_DOUBLES


def whole_level_fits_one_wave(ny, nx):
    """True when that SSOR runs on the barrier-free single-wave kernel (at most two rows per lane)."""
    return whole_level_fits_lds(ny, nx) and ny - 1 <= 128


def choose_n_para(gny, gnx, world=1, at_least=1):
    """Smallest number of Jacobi levels >= at_least that make_plan accepts for this grid and rank
    count and whose first whole-domain level runs on the fastest SSOR kernel (LDS-resident, one wave).  On a grid that grows with the rank count
    (weak scaling over rows) that level grows too, so more ranks want more Jacobi levels."""
    n_para = at_least
    while True:
        try:
            plan = make_plan(gny, gnx, n_para, world, 0)
        except ValueError:
            return None
        if whole_level_fits_one_wave(plan.whole[0].ny, plan.whole[0].nx):
            return n_para
        n_para += 1


T_ROUND_DEFAULT = 110e-6    # one exchange round on the cycle's critical path (s), unless measured
GATHER_BW_DEFAULT = 100e9   # rate at which an all-gathered array arrives at a rank (B/s), unless measured


def choose_n_dist(gny, gnx, n_para, world, halo, t_round=None, gather_bw=None, stream_rate=4.5e12, bytes_per_point=36):
    """How many of the Jacobi levels (the finest ones) to cut into slabs, by a cost model of one V-cycle:
      * every distributed level costs two exchange rounds (t_round each: what a batch of row
        sends/receives adds to the cycle where no pass is long enough to hide it; the rows themselves
        are a few hundred KB);
      * the first replicated level's right-hand side is all-gathered: its bytes x (world-1)/world
        arrive at every rank at gather_bw (xGMI is point-to-point: a ring moves ~one link's worth);
      * every replicated level is computed by all ranks: its points x bytes_per_point at stream_rate.
    t_round / gather_bw: measured values (bench.py measures both at start-up); None = the environment's
    MG_T_ROUND_US / MG_GATHER_GBPS, else the defaults above.  Every rank must pass the same values.
    Returns the k in 1..n_para with the least cost among those make_plan accepts (slabs of at least `halo` rows,
    rows per rank divisible by 2**k)."""
    import os
    if t_round is None:
        t_round = float(os.environ["MG_T_ROUND_US"]) * 1e-6 if os.environ.get("MG_T_ROUND_US") else T_ROUND_DEFAULT
    if gather_bw is None:
        gather_bw = float(os.environ["MG_GATHER_GBPS"]) * 1e9 if os.environ.get("MG_GATHER_GBPS") else GATHER_BW_DEFAULT
    R0 = (gny - 1) // world
    best, best_cost = 1, None
    for k in range(1, n_para + 1):
        if (R0 >> (k - 1)) < max(halo, 1) or R0 % (1 << k):
            break  # make_plan would refuse it: slabs shorter than the halo, or a slab boundary on an odd row
        pts = lambda l: (((gny - 1) >> l) + 1) * (((gnx - 1) >> l) + 1)  # noqa: E731
        cost = 2 * k * t_round + 8 * pts(k) * (world - 1) / world / gather_bw
        cost += sum(pts(l) for l in range(k, n_para)) * bytes_per_point / stream_rate
        if best_cost is None or cost < best_cost:
            best, best_cost = k, cost
    return best
