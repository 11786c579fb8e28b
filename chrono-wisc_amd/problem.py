"""Python mirror of ChFsiProblemCartesian (src/chrono_fsi/sph/ChFsiProblemSPH.{h,cpp}): the set-up helper most of the
reference's demos go through -- SPH markers on an integer grid, box-container BCE markers, rigid bodies with embedded-marker
pruning, initial-property callbacks, computational domain from the marker AABB -- on top of the facade mirror (fsi.py) and the
Chrono-free rigid-body driver (rigid.py).  The reference keeps grid points in std::unordered_set, so its marker order is
unspecified; here markers are staged in grid order (Ix, Iy, Iz nested), which no result depends on."""
import numpy as np

from . import fsi, rigid


class BoxSide:  # ChFsiProblemSPH.h BoxSide flags
    NONE, X_POS, X_NEG, Y_POS, Y_NEG, Z_POS, Z_NEG = 0x0000, 1 << 0, 1 << 1, 1 << 2, 1 << 3, 1 << 4, 1 << 5
    ALL = 0xFFFF


def depth_pressure_properties(zero_height):
    """DepthPressurePropertiesCallback (ChFsiProblemSPH.h:480-495): (rho, p, mu, v) per point."""
    def cb(sysSPH, pts):
        gz = abs(sysSPH.GetGravitationalAcceleration()[2])
        c2 = float(sysSPH.GetSoundSpeed()) ** 2
        p0 = sysSPH.GetDensity() * gz * (zero_height - pts[:, 2])
        return np.stack([sysSPH.GetDensity() + p0 / c2, p0, np.full_like(p0, sysSPH.GetViscosity())], axis=1), np.zeros_like(pts)
    return cb


def read_pgm(path):
    """Gray-level image (binary P5 or ASCII P2 portable graymap) as (gray[ix, iy], maxval) with the top-left pixel at [0, 0]."""
    data = open(path, "rb").read()
    tok, pos = [], 0
    while len(tok) < 4:  # magic, width, height, maxval; '#' comments allowed between them
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        tok.append(data[pos:end])
        pos = end
    magic, w, h, maxval = tok[0], int(tok[1]), int(tok[2]), int(tok[3])
    if magic == b"P5":
        dt = np.dtype(">u2") if maxval > 255 else np.uint8
        img = np.frombuffer(data, dtype=dt, count=w * h, offset=pos + 1).reshape(h, w)
    elif magic == b"P2":
        img = np.array(data[pos:].split(), dtype=np.int64).reshape(h, w)
    else:
        raise ValueError("read_pgm: %s is not a P2/P5 graymap" % path)
    return np.ascontiguousarray(img.T.astype(np.int64)), (65535 if maxval > 255 else 255)


DepthPressurePropertiesCallback = depth_pressure_properties  # the reference's class name


class CylinderShape:
    """utils::ChBodyGeometry::CylinderShape: axis along the shape frame's z."""
    def __init__(self, pos, rot, radius, length):
        self.pos, self.rot, self.radius, self.length = np.asarray(pos, dtype=np.float64), rot, float(radius), float(length)

    def bce(self, sysSPH):  # CreateBCEFsiBody, sph/ChFsiFluidSystemSPH.cpp:997-1003
        return fsi.frame_transform(sysSPH.CreatePointsCylinderInterior(self.radius, self.length, True), self.pos, self.rot)

    def inside(self, p_loc, envelope):  # InsidePoint, ChFsiProblemSPH.cpp:324-333
        pp = (p_loc - self.pos) @ fsi.rotmat_from_quat(self.rot)
        return (pp[:, 0] ** 2 + pp[:, 1] ** 2 <= (self.radius + envelope) ** 2) & (np.abs(pp[:, 2]) <= self.length / 2 + envelope)


class BoxShape:
    def __init__(self, pos, rot, dims):
        self.pos, self.rot, self.dims = np.asarray(pos, dtype=np.float64), rot, np.asarray(dims, dtype=np.float64)

    def bce(self, sysSPH):
        return fsi.frame_transform(sysSPH.CreatePointsBoxInterior(tuple(self.dims)), self.pos, self.rot)

    def inside(self, p_loc, envelope):  # :315-322
        pp = (p_loc - self.pos) @ fsi.rotmat_from_quat(self.rot)
        h = self.dims / 2 + envelope
        return (np.abs(pp[:, 0]) <= h[0]) & (np.abs(pp[:, 1]) <= h[1]) & (np.abs(pp[:, 2]) <= h[2])


class SphereShape:
    def __init__(self, pos, radius):
        self.pos, self.radius = np.asarray(pos, dtype=np.float64), float(radius)

    def bce(self, sysSPH):
        return sysSPH.CreatePointsSphereInterior(self.radius, True) + self.pos

    def inside(self, p_loc, envelope):  # :308-313
        d = p_loc - self.pos
        return (d * d).sum(axis=1) <= (self.radius + envelope) ** 2


class ChFsiProblemCartesian:
    def __init__(self, spacing, api=None, device=-1):
        self.sysSPH = fsi.ChFsiFluidSystemSPH(api, device=device)
        self.spacing = float(spacing)
        self.bodies = []       # (rigid.RigidBody, [shapes], check_embedded)
        self.props_cb = None
        self.sph = np.zeros((0, 3), dtype=np.int64)
        self.bce = np.zeros((0, 3), dtype=np.int64)
        self.offset_sph = np.zeros(3)
        self.offset_bce = np.zeros(3)
        self.domain = None
        self.fsi_system = None
        self.pre_pressure_scale0 = 1.01  # ParticlePropertiesCallback::pre_pressure_scale0 (sph/ChFsiProblemSPH.h:120)

    # -- forwarding setters (ChFsiProblemSPH.h:66-120)
    def SetGravitationalAcceleration(self, g):
        self.sysSPH.SetGravitationalAcceleration(g)
        self._gravity = tuple(g)

    def SetStepSizeCFD(self, h):
        self.sysSPH.SetStepSize(h)

    def SetCfdSPH(self, props):
        self.sysSPH.SetCfdSPH(props)

    def SetSPHParameters(self, sp):
        self.sysSPH.SetSPHParameters(sp)

    def RegisterParticlePropertiesCallback(self, cb):
        self.props_cb = cb

    def SetComputationalDomain(self, cmin, cmax, bc=fsi.BC_NONE):
        self.domain = (tuple(cmin), tuple(cmax), bc)

    def AddRigidBody(self, body, shapes, check_embedded=True):
        """body: rigid.RigidBody; shapes: primitives in the body frame (ChFsiProblemSPH::AddRigidBody, cpp:106-113)."""
        self.bodies.append((body, list(shapes), check_embedded))
        return len(self.bodies) - 1

    # -- integer grid <-> physical coordinates relative to the offset (ChFsiProblemCartesian::Snap2Grid / Grid2Point, cpp:1049-1059)
    def Snap2Grid(self, points):
        return np.floor(np.asarray(points, dtype=np.float64) / self.spacing + 0.5).astype(np.int64)

    def Grid2Point(self, grid):
        return self.spacing * np.asarray(grid, dtype=np.float64).reshape(-1, 3)

    # -- Construct (cpp:712-749) + AddBoxContainer (:937-1047)
    def Construct(self, box_size, pos, side_flags):
        d0 = self.spacing
        rnd = lambda v: int(np.floor(v + 0.5))
        self.N = Nx, Ny, Nz = rnd(box_size[0] / d0) + 1, rnd(box_size[1] / d0) + 1, rnd(box_size[2] / d0) + 1
        Ix, Iy, Iz = np.meshgrid(np.arange(Nx), np.arange(Ny), np.arange(Nz), indexing="ij")
        self.sph = np.stack([Ix.ravel(), Iy.ravel(), Iz.ravel()], axis=1)
        self.offset_sph = np.asarray(pos, dtype=np.float64) - np.array([box_size[0] / 2, box_size[1] / 2, 0.0])
        if side_flags != BoxSide.NONE:
            self.AddBoxContainer(box_size, pos, side_flags)

    # -- Construct(heightmap_file, length, width, height_range, depth, uniform_depth, pos, side_flags) (cpp:751-935): a terrain patch
    # whose surface follows a gray-level image (black -> height_range[0], white -> height_range[1], bi-linear between pixels),
    # filled with SPH markers `depth` deep below the surface (uniform_depth) or down to a flat bottom, BCE layers underneath and
    # on the chosen sides.  `heightmap`: a 2-D integer array gray[ix, iy] (image columns = x, image rows = y, top-left origin like
    # the reference's STB reader) with `gray_range` = 255 or 65535, or the name of a binary/ASCII PGM file.
    def ConstructFromHeightmap(self, heightmap, length, width, height_range, depth, uniform_depth, pos, side_flags, gray_range=None):
        if isinstance(heightmap, str):
            gmap, gray_range = read_pgm(heightmap)
        else:
            gmap = np.asarray(heightmap)
            if gray_range is None:
                gray_range = 65535 if gmap.dtype.itemsize > 1 else 255
        gmap = gmap.astype(np.float64)
        nx, ny = gmap.shape
        d0 = self.spacing
        L = int(self.sysSPH.GetNumBCELayers())
        rnd = lambda v: np.where(np.asarray(v) >= 0, np.floor(np.asarray(v) + 0.5), -np.floor(-np.asarray(v) + 0.5)).astype(np.int64)
        dx, dy = length / (nx - 1), width / (ny - 1)
        h_scale = (height_range[1] - height_range[0]) / gray_range
        Iz_min = int(rnd((height_range[0] + gmap.min() * h_scale) / d0))
        Iz_max = int(rnd((height_range[0] + gmap.max() * h_scale) / d0))
        Nx, Ny = int(rnd(length / d0)), int(rnd(width / d0))
        Dx, Dy = length / Nx, width / Ny
        Nx, Ny = Nx + 1, Ny + 1
        Nz = int(rnd(depth / d0)) + 1
        x, y = np.arange(Nx) * Dx, np.arange(Ny) * Dy
        ix1, ix2 = np.floor(x / dx).astype(np.int64), np.ceil(x / dx).astype(np.int64)
        ix1[-1], ix2[0] = nx - 2, 1
        iy1, iy2 = np.floor(y / dy).astype(np.int64), np.ceil(y / dy).astype(np.int64)
        iy1[-1], iy2[0] = ny - 2, 1
        wx, wy = ((ix2 * dx - x) / dx)[:, None], ((iy2 * dy - y) / dy)[None, :]
        h1 = wx * gmap[np.ix_(ix1, iy1)] + (1 - wx) * gmap[np.ix_(ix2, iy1)]
        h2 = wx * gmap[np.ix_(ix1, iy2)] + (1 - wx) * gmap[np.ix_(ix2, iy2)]
        Iz = rnd((height_range[0] + (wy * h1 + (1 - wy) * h2) * h_scale) / d0)      # surface level per column, [Nx, Ny]
        nz = np.full_like(Iz, Nz) if uniform_depth else np.maximum(Iz + Nz, 0)
        IX, IY = np.meshgrid(np.arange(Nx), np.arange(Ny), indexing="ij")
        cols = np.stack([IX.ravel(), IY.ravel()], axis=1)
        top, cnt = Iz.ravel(), nz.ravel()

        def stack(first_level, count):  # per column: levels first_level, first_level - 1, ... (count of them)
            tot = int(count.sum())
            rep = np.repeat(np.arange(len(count)), count)
            k = np.arange(tot) - np.repeat(np.cumsum(count) - count, count)
            return np.stack([cols[rep, 0], cols[rep, 1], first_level[rep] - k], axis=1)
        sph = stack(top, cnt)
        B = []
        if side_flags & BoxSide.Z_NEG:
            B.append(stack(top - cnt, np.full_like(cnt, L)))
        nz_min, nz_max = Iz_min - Nz - L, Iz_max + L
        a = np.arange

        def block(ix, iy):
            X, Y, Z = np.meshgrid(ix, iy, a(nz_min, nz_max), indexing="ij")
            B.append(np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1))
        if side_flags & BoxSide.X_NEG: block(a(-L, 0), a(Ny))
        if side_flags & BoxSide.X_POS: block(Nx - 1 - a(-L, 0), a(Ny))
        if side_flags & BoxSide.Y_NEG: block(a(-L, Nx + L), a(-L, 0))
        if side_flags & BoxSide.Y_POS: block(a(-L, Nx + L), Ny - 1 - a(-L, 0))
        # pixels start at the top-left corner: flip y so that the markers start at the bottom-left (cpp:912-922)
        sph[:, 1] = (Ny - 1) - sph[:, 1]
        _, first = np.unique(sph, axis=0, return_index=True)
        self.sph = sph[np.sort(first)]
        if B:
            bce = np.concatenate(B, axis=0)
            bce[:, 1] = (Ny - 1) - bce[:, 1]
            bce = np.concatenate([self.bce, bce], axis=0)
            _, first = np.unique(bce, axis=0, return_index=True)
            self.bce = bce[np.sort(first)]
        self.offset_sph = np.asarray(pos, dtype=np.float64) - np.array([length / 2, width / 2, 0.0])
        if side_flags != BoxSide.NONE:
            self.offset_bce = self.offset_sph.copy()
        self.N = (Nx, Ny, Nz)

    # -- Construct(sph_file, bce_file, pos, use_grid_coordinates) (cpp:646-711): marker sets from two text files, one marker
    # per line, "x y z" or "x,y,z"; integer grid coordinates, or physical positions rounded to the grid (std::round: half away
    # from zero).  Lines that do not parse are skipped, duplicates collapse (the reference inserts into a set).
    def ConstructFromFiles(self, sph_file, bce_file, pos, use_grid_coordinates=True):
        def read(path):
            pts = []
            with open(path) as f:
                for line in f:
                    tok = line.replace(",", " ").split()
                    try:
                        if use_grid_coordinates:
                            # operator>> into int: leading integer part of each token
                            v = [int(tok[k]) for k in range(3)]
                        else:
                            x = [float(tok[k]) / self.spacing for k in range(3)]
                            v = [int(np.sign(a) * np.floor(abs(a) + 0.5)) for a in x]
                    except (ValueError, IndexError):
                        continue
                    pts.append(v)
            a = np.asarray(pts, dtype=np.int64).reshape(-1, 3)
            if len(a):
                _, first = np.unique(a, axis=0, return_index=True)
                a = a[np.sort(first)]
            return a
        self.sph = read(sph_file)
        self.bce = read(bce_file)
        self.offset_sph = np.asarray(pos, dtype=np.float64)
        self.offset_bce = self.offset_sph.copy()
        return len(self.sph), len(self.bce)

    def SaveGridFiles(self, sph_file, bce_file):
        """Write the current marker sets in the grid-coordinate format ConstructFromFiles reads (a restart path for scenes
        built from boxes and containers; the reference only reads this format)."""
        for path, pts, off in ((sph_file, self.sph, self.offset_sph), (bce_file, self.bce, self.offset_bce)):
            shift = np.floor((off - self.offset_sph) / self.spacing + 0.5).astype(np.int64)
            with open(path, "w") as f:
                for x, y, z in pts + shift:
                    f.write("%d, %d, %d\n" % (x, y, z))
        return self.offset_sph.copy()

    def AddBoxContainer(self, box_size, pos, side_flags):
        d0 = self.spacing
        L = int(self.sysSPH.GetNumBCELayers())
        rnd = lambda v: int(np.floor(v + 0.5))
        Nx, Ny, Nz = rnd(box_size[0] / d0) + 1, rnd(box_size[1] / d0) + 1, rnd(box_size[2] / d0) + 1
        a, B = np.arange, []

        def block(ix, iy, iz):
            X, Y, Z = np.meshgrid(ix, iy, iz, indexing="ij")
            B.append(np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1))
        if side_flags & BoxSide.Z_NEG: block(a(Nx), a(Ny), -a(1, L + 1))
        if side_flags & BoxSide.Z_POS: block(a(Nx), a(Ny), Nz - 1 + a(1, L + 1))
        if side_flags & BoxSide.X_NEG: block(a(-L, 0), a(Ny), a(-L, Nz + L))
        if side_flags & BoxSide.X_POS: block(Nx - 1 - a(-L, 0), a(Ny), a(-L, Nz + L))
        if side_flags & BoxSide.Y_NEG: block(a(-L, Nx + L), a(-L, 0), a(-L, Nz + L))
        if side_flags & BoxSide.Y_POS: block(a(-L, Nx + L), Ny - 1 - a(-L, 0), a(-L, Nz + L))
        Bg = np.concatenate([self.bce] + B, axis=0) if B else self.bce
        _, first = np.unique(Bg, axis=0, return_index=True)  # the reference inserts into a set
        self.bce = Bg[np.sort(first)]
        self.offset_bce = np.asarray(pos, dtype=np.float64) - np.array([box_size[0] / 2, box_size[1] / 2, 0.0])
        return len(self.bce)

    # -- Initialize (cpp:185-298): prune embedded markers, create markers, domain, initialize the fluid system
    def Initialize(self):
        s, d0 = self.sysSPH, self.spacing
        keep = np.ones(len(self.sph), dtype=bool)
        body_bce = []
        for body, shapes, check_embedded in self.bodies:
            local = np.concatenate([sh.bce(s) for sh in shapes], axis=0)
            body_bce.append(local)
            if not check_embedded or not len(self.sph):
                continue
            bce_abs = fsi.frame_transform(local, body.pos, body.rot)
            snapped = self.Snap2Grid(bce_abs - self.offset_sph)
            lo, hi = snapped.min(axis=0), snapped.max(axis=0)
            cand = np.all((self.sph >= lo) & (self.sph <= hi), axis=1) & keep
            if cand.any():
                p_abs = self.Grid2Point(self.sph[cand]) + self.offset_sph
                p_loc = (p_abs - body.pos) @ fsi.rotmat_from_quat(body.rot)  # TransformPointParentToLocal
                inside = np.zeros(len(p_loc), dtype=bool)
                for sh in shapes:
                    inside |= sh.inside(p_loc, d0)
                idx = np.nonzero(cand)[0]
                keep[idx[inside]] = False
        sph_pts = self.Grid2Point(self.sph[keep]) + self.offset_sph
        bce_pts = self.Grid2Point(self.bce) + self.offset_bce
        if self.props_cb is not None:
            rpm, vel = self.props_cb(s, sph_pts)
        else:
            rpm = np.tile(np.array([[s.GetDensity(), 0.0, s.GetViscosity()]]), (len(sph_pts), 1))
            vel = np.zeros_like(sph_pts)
        if s.GetPhysicsProblemString() == "CRM":
            # cpp:253-263: isotropic initial stress -p0 and, for the MCC rheology, the consolidation pressure p0 * pre_pressure_scale0
            p0 = np.asarray(rpm)[:, 1]
            s.AddSPHParticles(sph_pts, rpm, vel, tau_diag=-np.stack([p0, p0, p0], axis=1), tau_offdiag=np.zeros((len(p0), 3)), pc=p0 * self.pre_pressure_scale0)
        else:
            s.AddSPHParticles(sph_pts, rpm, vel)
        if len(bce_pts):
            s.AddBCEBoundary(bce_pts)
        for (body, _sh, _ce), local in zip(self.bodies, body_bce):
            s.AddFsiBody(local, body.state())
        if self.domain is not None:
            s.SetComputationalDomain(*self.domain)
        else:
            allp = np.concatenate([sph_pts, bce_pts], axis=0)
            L = int(s.GetNumBCELayers())
            s.SetComputationalDomain(tuple(allp.min(axis=0) - L * d0), tuple(allp.max(axis=0) + L * d0), fsi.BC_NONE)
        s.Initialize()
        self.fsi_system = rigid.FsiSystem(s, [b for b, _s, _c in self.bodies])
        return self.fsi_system

    def DoStepDynamics(self, step):
        self.fsi_system.DoStepDynamics(step)


class CylSide:  # ChFsiDefinitionsSPH.h:133-135
    NONE, SIDE_INT, SIDE_EXT, Z_NEG, Z_POS, ALL = 0, 1, 2, 4, 8, 0xFFFF


class ChFsiProblemCylindrical(ChFsiProblemCartesian):
    """ChFsiProblemCylindrical (sph/ChFsiProblemSPH.cpp:1287-1465): markers on the polar grid (Ir, Ia, Iz) -- ring Ir holds
    floor(2 pi Ir) points -- for a cylindrical or annular fluid volume with optional container walls.  Everything else
    (embedded-body pruning, property callback, domain, Initialize) is the base class's, through Snap2Grid / Grid2Point."""

    @staticmethod
    def _ring(Ir, use_round=False):
        return int(np.floor(2 * np.pi * Ir + 0.5)) if use_round else int(np.floor(2 * np.pi * Ir))

    def _shells(self, rings, iz, use_round=False):
        out = []
        for Ir in rings:
            Na = self._ring(Ir, use_round)
            if Na <= 0 or len(iz) == 0:
                continue
            Ia, Iz = np.meshgrid(np.arange(Na), iz, indexing="ij")
            out.append(np.stack([np.full(Ia.size, Ir), Ia.ravel(), Iz.ravel()], axis=1))
        return out

    def Construct(self, radius_inner, radius_outer, height, pos, side_flags):  # cpp:1289-1334
        d0 = self.spacing
        rnd = lambda v: int(np.floor(v + 0.5))
        filled = radius_inner < 0.5 * d0
        Ir_start, Nr, Nz = rnd(radius_inner / d0), rnd((radius_outer - radius_inner) / d0) + 1, rnd(height / d0) + 1
        parts = []
        if filled:
            parts.append(np.stack([np.zeros(Nz, dtype=np.int64), np.zeros(Nz, dtype=np.int64), np.arange(Nz)], axis=1))
        parts += self._shells(range(Ir_start, Ir_start + Nr), np.arange(Nz))
        self.sph = self._unique(parts)
        self.offset_sph = np.asarray(pos, dtype=np.float64)
        if side_flags != CylSide.NONE:
            self.AddCylindricalContainer(radius_inner, radius_outer, height, pos, side_flags)

    @staticmethod
    def _unique(parts):
        if not parts:
            return np.zeros((0, 3), dtype=np.int64)
        G = np.concatenate(parts, axis=0).astype(np.int64)
        _, first = np.unique(G, axis=0, return_index=True)  # the reference inserts into a set
        return G[np.sort(first)]

    def AddCylindricalContainer(self, radius_inner, radius_outer, height, pos, side_flags):  # cpp:1336-1432
        d0 = self.spacing
        L = int(self.sysSPH.GetNumBCELayers())
        rnd = lambda v: int(np.floor(v + 0.5))
        filled = radius_inner < 0.5 * d0
        Ir_start, Nr, Nz = rnd(radius_inner / d0), rnd((radius_outer - radius_inner) / d0) + 1, rnd(height / d0) + 1
        col = lambda iz: np.stack([np.zeros(len(iz), dtype=np.int64), np.zeros(len(iz), dtype=np.int64), iz], axis=1)
        parts = [self.bce]
        if side_flags & CylSide.Z_NEG:
            if filled:
                parts.append(col(-np.arange(1, L + 1)))
            parts += self._shells(range(Ir_start, Ir_start + Nr), -np.arange(1, L + 1))
        if side_flags & CylSide.Z_POS:
            if filled:
                parts.append(col(np.arange(Nz, Nz + L)))
            parts += self._shells(range(Ir_start, Ir_start + Nr), np.arange(Nz, Nz + L))
        if side_flags & CylSide.SIDE_EXT:  # rings Ir_start+Nr .. Ir_start+Nr+L (inclusive), round(2 pi Ir) points each
            parts += self._shells(range(Ir_start + Nr, Ir_start + Nr + L + 1), np.arange(-L, Nz + L), use_round=True)
        if (side_flags & CylSide.SIDE_INT) and not filled:
            if Ir_start <= L:
                parts.append(col(np.arange(-L, Nz + L)))
            parts += self._shells([Ir for Ir in range(Ir_start - L, Ir_start) if Ir > 0], np.arange(-L, Nz + L), use_round=True)
        self.bce = self._unique(parts)
        self.offset_bce = np.asarray(pos, dtype=np.float64)
        return len(self.bce)

    def Snap2Grid(self, points):  # cpp:1434-1446
        p = np.asarray(points, dtype=np.float64).reshape(-1, 3)
        r = np.sqrt(p[:, 0] ** 2 + p[:, 1] ** 2)
        theta = np.arctan2(p[:, 1], p[:, 0])
        theta = np.where(theta < 0, theta + 2 * np.pi, theta)
        rnd = lambda v: np.where(v >= 0, np.floor(v + 0.5), -np.floor(-v + 0.5))
        ir = rnd(r / self.spacing).astype(np.int64)
        with np.errstate(divide="ignore", invalid="ignore"):
            d_theta = 2 * np.pi / np.floor(2 * np.pi * ir)
            ia = np.where(ir > 0, rnd(theta / d_theta), 0)
        return np.stack([ir, ia.astype(np.int64), rnd(p[:, 2] / self.spacing).astype(np.int64)], axis=1)

    def Grid2Point(self, grid):  # cpp:1448-1463
        g = np.asarray(grid, dtype=np.int64).reshape(-1, 3)
        r = g[:, 0] * self.spacing
        with np.errstate(divide="ignore", invalid="ignore"):
            theta = np.where(g[:, 0] > 0, g[:, 1] * (2 * np.pi / np.floor(2 * np.pi * np.maximum(g[:, 0], 1))), 0.0)
        return np.stack([np.where(g[:, 0] > 0, r * np.cos(theta), 0.0), np.where(g[:, 0] > 0, r * np.sin(theta), 0.0), g[:, 2] * self.spacing], axis=1)


class WavemakerType:  # ChFsiProblemWavetank::WavemakerType
    PISTON, FLAP = 0, 1


class WaveTankRampBeach:
    """ChFsiProblemSPH.h:505-522: flat bottom up to x_start, then a ramp of slope alpha."""

    def __init__(self, x_start, alpha):
        self.x_start, self.alpha = x_start, alpha

    def __call__(self, x):
        return 0.0 if x <= self.x_start else self.alpha * (x - self.x_start)


class WaveTankParabolicBeach:
    """ChFsiProblemSPH.h:526-543: flat bottom up to x_start, then alpha * sqrt(x - x_start)."""

    def __init__(self, x_start, alpha):
        self.x_start, self.alpha = x_start, alpha

    def __call__(self, x):
        return 0.0 if x <= self.x_start else self.alpha * np.sqrt(x - self.x_start)


class ChFsiProblemWavetank(ChFsiProblemCartesian):
    """ChFsiProblemWavetank (sph/ChFsiProblemSPH.cpp:1063-1285): a tank with an optional bottom profile (beach), lateral walls or
    lateral periodicity, an end wall, and a wavemaker -- a plate moved by a prescribed actuation function (piston: translation
    along x; flap: rotation about the y axis at the tank bottom), which is an FSI body of the fluid system."""

    def __init__(self, spacing, api=None, device=-1):
        super().__init__(spacing, api, device)
        self.profile, self.end_wall, self.periodic_BC = None, True, False
        self.wavemaker_body = None

    def SetProfile(self, profile, end_wall):
        self.profile, self.end_wall = profile, end_wall

    def SetLateralPeriodicBC(self, periodic_BC):
        self.periodic_BC = bool(periodic_BC)

    def ConstructWaveTank(self, wavemaker_type, pos, box_size, depth, actuation):
        d0 = self.spacing
        L = int(self.sysSPH.GetNumBCELayers())
        rnd = lambda v: int(np.floor(v + 0.5)) if v >= 0 else -int(np.floor(-v + 0.5))
        Nx, Ny = rnd(box_size[0] / d0) + 1, rnd(box_size[1] / d0) + 1
        Nzc, Nzf = rnd(box_size[2] / d0) + 1, rnd(depth / d0) + 1
        Iy0 = 0 if self.periodic_BC else L
        sph, bce = [], []
        a = np.arange

        def block(dst, ix, iy, iz):
            if len(ix) and len(iy) and len(iz):
                X, Y, Z = np.meshgrid(ix, iy, iz, indexing="ij")
                dst.append(np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1))
        Iz0 = 0
        for Ix in range(Nx):
            z = self.profile(Ix * d0) if self.profile else 0.0
            Iz0 = rnd(z / d0)
            block(sph, [Ix], a(Ny), a(Iz0, Nzf))                       # SPH particles above the bottom height
            block(bce, [Ix], a(Ny), Iz0 - a(1, L + 1))                 # BCE markers below it
            side = a(Iz0 - L, Nzc + L)
            block(bce, [Ix], a(-Iy0, 0), side)                         # lateral walls (none under lateral periodicity)
            block(bce, [Ix], Ny - 1 - a(-Iy0, 0), side)
        for Ix in range(-L, 0):                                        # around the wavemaker body (negative Ix)
            block(bce, [Ix], a(-Iy0, Ny + Iy0), -a(1, L + 1))
            block(bce, [Ix], a(-Iy0, 0), a(-L, Nzc + L))
            block(bce, [Ix], Ny - 1 - a(-Iy0, 0), a(-L, Nzc + L))
        if not self.profile:
            self.end_wall = True
        if self.end_wall:
            block(bce, a(Nx, Nx + L), a(-Iy0, Ny + Iy0), a(Iz0 - L, Nzc + L))
        uniq = lambda parts: (lambda G: G[np.sort(np.unique(G, axis=0, return_index=True)[1])])(np.concatenate(parts, axis=0).astype(np.int64))
        self.sph, self.bce = uniq(sph), uniq(bce)
        self.N = (Nx, Ny, Nzf)
        pos = np.asarray(pos, dtype=np.float64)
        self.offset_sph = pos - np.array([box_size[0] / 2, box_size[1] / 2, 0.0])
        self.offset_bce = self.offset_sph.copy()
        if self.periodic_BC:
            lo = np.array([-L * d0, (0 + 0.5) * d0, -L * d0])
            hi = np.array([(Nx + L) * d0, ((Ny - 1) - 0.5) * d0, (Nzc + L) * d0])
            self.domain = (tuple(self.offset_sph + lo), tuple(self.offset_sph + hi), fsi.BC_Y_PERIODIC)
        thickness = (L - 1) * d0
        size = np.array([thickness, box_size[1], box_size[2]])
        wpos = np.array([-box_size[0] / 2 - thickness / 2 - d0, 0.0, box_size[2] / 2])
        if wavemaker_type == WavemakerType.PISTON:
            body = rigid.PrescribedBody(pos + wpos, actuation, "piston")
        else:
            pivot = pos + wpos - np.array([0.0, 0.0, box_size[2] / 2])
            wpos = wpos - np.array([0.0, 0.0, thickness / 2])
            size = size + np.array([0.0, 0.0, thickness])
            body = rigid.PrescribedBody(pos + wpos, actuation, "flap", pivot=pivot)
        self.wavemaker_body, self.wavemaker_size = body, size
        return body

    def Initialize(self):  # cpp:1267-1285: the wavemaker plate becomes an FSI body (box geometry), then the base initialization
        if self.wavemaker_body is not None and not any(b is self.wavemaker_body for b, _s, _c in self.bodies):
            self.AddRigidBody(self.wavemaker_body, [BoxShape((0.0, 0.0, 0.0), fsi.QUNIT, tuple(self.wavemaker_size))], True)
        return super().Initialize()
