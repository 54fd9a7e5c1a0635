# CEGHip.jl -- reference-side binding of libceg_hip.so (include/ceg_hip.h).
#
# Drop-in for the two loop nests of CrystalEnergyGrids.jl that fill an energy grid:
#
#     create_grid_vdw      src/grids.jl:137-157   (loop nest :144-150)
#     create_grid_coulomb  src/grids.jl:159-185   (loop nest :171-177)
#
# Everything else -- ProbeSystem, GridCoordinatesSetup, initialize_ewald, the unit constants,
# `_create_grid_common` and the file layout -- is the reference's own code, called unchanged, so
# `retrieve_or_create_grid` (src/raspa.jl:420-439), `setup_RASPA`, `setup_montecarlo`,
# `CrystalEnergySetup` and `energy_point` keep working without modification.
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no Julia toolchain.  The Python
# package `ceg_hip` is the tested twin of this file (same argument marshalling, same C calls), and
# tests/test_boundary_static.py parses every `ccall` below and checks symbol, arity and C types
# against include/ceg_hip.h and ceg_hip/_abi.py.
#
# Usage (after `using CrystalEnergyGrids`):
#     include("CEGHip.jl"); CEGHip.install!()        # overrides the two methods
#     ENV["CEG_HIP_LIB"] = "/path/to/libceg_hip.so"   # optional, default next to this file
#     ENV["CEG_HIP_NGPUS"] = "8"                      # optional, default 1
module CEGHip

import CrystalEnergyGrids as CEG
using CrystalEnergyGrids: ProbeSystem, ForceField, InteractionRule, InteractionRuleSum, FF,
                          EwaldFramework, GridCoordinatesSetup, GRID_TO_KELVIN,
                          COULOMBIC_CONVERSION_FACTOR, TÅ
using Unitful, UnitfulAtomic
using AtomsBase: AbstractSystem
using StaticArrays

const LIB = Ref(get(ENV, "CEG_HIP_LIB", joinpath(@__DIR__, "..", "csrc", "libceg_hip.so")))
ngpus() = parse(Int, get(ENV, "CEG_HIP_NGPUS", "1"))

# struct ceg_rule { int32 kind; int32 _pad; double p[3]; double shift; }   (include/ceg_hip.h)
struct CegRule
    kind::Int32
    _pad::Int32
    p1::Float64
    p2::Float64
    p3::Float64
    shift::Float64
end

_rules(r::InteractionRule) = (r,)
_rules(r::InteractionRuleSum) = r.rules

"Flatten `ff.interactions[:, probe]` (what `derivatives_nocutoff`, src/forcefields.jl:302-304, dispatches on)."
function rule_table(ff::ForceField, probe::Int)
    n = size(ff.interactions, 1)
    flat = CegRule[]
    offsets = Int32[0]
    for k in 1:n
        for r in _rules(ff.interactions[k, probe])
            p = r.params
            push!(flat, CegRule(Int32(Int(r.kind)), 0, get(p, 1, 0.0), get(p, 2, 0.0), get(p, 3, 0.0), r.shift))
        end
        push!(offsets, Int32(length(flat)))
    end
    flat, offsets
end

"Raise what `derivativesGrid` (src/interactions.jl:442-443,462-467) would raise for the kinds present."
function check_rules(ff::ForceField, probe::Int, kinds)
    for k in unique(kinds), r in _rules(ff.interactions[k, probe])
        r.kind === FF.UndefinedInteraction && throw(CEG.UndefinedInteractionError())
        r.kind === FF.Coulomb && error("Coulomb interactions should not be taken into account in VdW grids.")
        r.kind === FF.Monomial && error("VdW grid not implemented for Monomial")
        r.kind === FF.Exponential && error("VdW grid not implemented for Exponential")
    end
end

function _check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:ceg_last_error, LIB[]), Cstring, ()))
    error("libceg_hip error $rc: $msg")
end

function _geometry(cset::GridCoordinatesSetup)
    dims = Int32[cset.dims...]
    size = Float64[NoUnits(x/u"Å") for x in cset.size]
    shift = Float64[NoUnits(x/u"Å") for x in cset.shift]
    Δ = Float64[NoUnits(x/u"Å") for x in cset.Δ]
    dims, size, shift, Δ
end

function _flatpos(probe::ProbeSystem)
    pos = Vector{Float64}(undef, 3*length(probe.positions))
    for (i, p) in enumerate(probe.positions)
        pos[3i-2] = p[1]; pos[3i-1] = p[2]; pos[3i] = p[3]
    end
    pos
end

"GPU replacement of the loop nest src/grids.jl:144-150; fills and returns `grid`."
function fill_grid_vdw!(grid::Array{Cfloat,4}, probe::ProbeSystem, cset::GridCoordinatesSetup, λ, thr)
    ff = probe.forcefield
    check_rules(ff, probe.probe, probe.atomkinds)
    rules, offsets = rule_table(ff, probe.probe)
    _, ortho, safemin = CEG.prepare_periodic_distance_computations(probe.mat)   # src/utils.jl:146-155
    cutoff2 = NoUnits(ff.cutoff^2/u"Å^2")                                          # src/probes.jl:75
    dims, size, shift, Δ = _geometry(cset)
    pos = _flatpos(probe)
    kinds = Int64.(probe.atomkinds)
    mat = Vector{Float64}(vec(probe.mat)); invmat = Vector{Float64}(vec(probe.invmat))   # column-major
    GC.@preserve grid pos kinds mat invmat rules offsets dims size shift Δ begin
        _check(ccall((:ceg_grid_vdw, LIB[]), Cint,
            (Ptr{Float64}, Ptr{Int64}, Int64, Ptr{Float64}, Ptr{Float64}, Int32, Float64, Float64,
             Ptr{CegRule}, Ptr{Int32}, Int32, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Float64, Float64, Ptr{Cfloat}, Int32),
            pos, kinds, length(kinds), mat, invmat, ortho, safemin^2, cutoff2,
            rules, offsets, length(offsets)-1, dims, size, shift, Δ, λ, thr, grid, ngpus()))
    end
    grid
end

"GPU replacement of the loop nest src/grids.jl:171-177."
function fill_grid_coulomb!(grid::Array{Cfloat,4}, probe::ProbeSystem, ewald::EwaldFramework, cset::GridCoordinatesSetup, λ, thr)
    _, ortho, safemin = CEG.prepare_periodic_distance_computations(probe.mat)
    cutoff2 = NoUnits(probe.forcefield.cutoff^2/u"Å^2")                            # src/probes.jl:98
    dims, size, shift, Δ = _geometry(cset)
    pos = _flatpos(probe)
    mat = Vector{Float64}(vec(probe.mat)); invmat = Vector{Float64}(vec(probe.invmat))
    α = NoUnits(ewald.α*u"Å")
    GC.@preserve grid pos mat invmat dims size shift Δ begin
        _check(ccall((:ceg_grid_coulomb, LIB[]), Cint,
            (Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Int32, Float64, Float64, Float64,
             Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Ptr{Cfloat}, Int32),
            pos, probe.charges, length(probe.charges), mat, invmat, ortho, safemin^2, cutoff2, α,
            dims, size, shift, Δ, λ, thr, grid, ngpus()))
    end
    grid
end

# Result arrays.  By default (round 4) the grid array is page-locked memory of the library (ceg_host_grid_alloc) wrapped as an Array: the
# build then copies every chunk D2H straight to its place (256^3 VdW grid: 10.5 instead of 13.4 ms, the driver-run figures are in
# BENCH_r04.json `oneshot`).  The Array goes back to the library's cache when it is finalized, or at once through `release!(grid)` when the
# caller is done with it (the reference's own caller, retrieve_or_create_grid, src/raspa.jl:420-439, drops the returned array and parses
# the file it has just written).  The page-locked memory in Julia's hands is BOUNDED by the library (CEG_HIP_PINNED_LIMIT_MB, default
# 4096 MB): beyond it ceg_host_grid_alloc refuses and an ordinary Array is used, so a collector that has not run yet cannot pin host
# memory without limit.  ENV["CEG_HIP_PINNED_RESULT"] = "0" turns the page-locked arrays off.
pinned_results() = get(ENV, "CEG_HIP_PINNED_RESULT", "1") != "0"

_free_pinned(g) = (ccall((:ceg_host_grid_free, LIB[]), Cint, (Ptr{Cfloat},), pointer(g)); nothing)

function result_array(cset::GridCoordinatesSetup)
    dims = (cset.dims[3]+1, cset.dims[2]+1, cset.dims[1]+1, 8)
    pinned_results() || return Array{Cfloat,4}(undef, dims...)
    d = Int32[cset.dims...]
    ptr = GC.@preserve d ccall((:ceg_host_grid_alloc, LIB[]), Ptr{Cfloat}, (Ptr{Int32},), d)
    ptr == C_NULL && return Array{Cfloat,4}(undef, dims...)      # over the limit (or no page-locked memory): the ordinary route
    grid = unsafe_wrap(Array, ptr, dims; own=false)
    finalizer(_free_pinned, grid)
    grid
end

"""
    release!(grid)

Hand the page-locked memory behind a grid returned by `create_grid_vdw` / `create_grid_coulomb` back to the library NOW instead of at
the next garbage collection.  `grid` must not be used afterwards.  A no-op for ordinary arrays.
"""
release!(grid::Array{Cfloat,4}) = (finalize(grid); nothing)     # runs (and retires) the finalizer registered by result_array

# The two methods below are the reference's (src/grids.jl:137-185) with the `@threads` loop nest
# replaced by one call; every other line is unchanged.
function create_grid_vdw(file, framework::AbstractSystem{3}, forcefield::ForceField, spacing::TÅ, atom::Symbol)
    cset, num_unitcell = CEG._setup_grid_common(framework, spacing, forcefield.cutoff)
    grid = result_array(cset)            # reference: Array{Cfloat,4}(undef, cset.dims[3]+1, cset.dims[2]+1, cset.dims[1]+1, 8)
    probe_vdw = ProbeSystem(framework, forcefield, atom)
    λ⁻¹ = GRID_TO_KELVIN
    λ = inv(λ⁻¹)
    fill_grid_vdw!(grid, probe_vdw, cset, λ, λ⁻¹*1e7)
    open(file, "w") do f
        CEG._create_grid_common(f, cset, num_unitcell)
        write(f, grid)
        write(f, NoUnits.(cset.cell.mat./u"Å"))
    end
    grid
end

function create_grid_coulomb(file, framework::AbstractSystem{3}, forcefield::ForceField, spacing::TÅ, _ewald=nothing)
    cset, num_unitcell = CEG._setup_grid_common(framework, spacing, 12.0u"Å")
    ewald = _ewald isa EwaldFramework ? _ewald : CEG.initialize_ewald(framework, num_unitcell)
    grid = result_array(cset)
    probe_coulomb = ProbeSystem(framework, forcefield)
    λ = ustrip(u"K*Å/e_au^2", COULOMBIC_CONVERSION_FACTOR)/GRID_TO_KELVIN
    λ⁻¹e7 = inv(λ)*1e7
    fill_grid_coulomb!(grid, probe_coulomb, ewald, cset, λ, λ⁻¹e7)
    open(file, "w") do f
        CEG._create_grid_common(f, cset, num_unitcell)
        write(f, ewald.precision)
        write(f, grid)
        write(f, NoUnits.(cset.cell.mat./u"Å"))
    end
    grid
end

# ------------------------------------------------------------------ all the grids of a setup in one pass
# setup_RASPA (src/raspa.jl:497-520) calls retrieve_or_create_grid once for the Coulomb grid and once per distinct atom of the
# molecule; each call that has to CREATE its grid runs a full pass over the framework.  ceg_grids_multi (include/ceg_hip.h)
# builds the VdW grids of up to 4 probes OF ANY RULE CLASS (round 4: Na + the C and O of CO2) and the Coulomb grid from one
# lattice-image list: the Lennard-Jones-only probes share accumulating loops, a Buckingham cation is launched alone or fused with the
# Coulomb grid.  (These multi-probe paths -- create_grids_multi, prebuild_grids!, result_array, interp_handle_from_file -- have never been
# executed: no Julia in the build image; tests/test_boundary_static.py checks ccall names and arities only.)

"True when `atom` meets every kind present in `probe` with at most one Lennard-Jones rule (such probes share accumulating loops; informational)."
function lj_only(ff::ForceField, probe::Int, kinds)
    for k in unique(kinds)
        real = [r for r in _rules(ff.interactions[k, probe]) if r.kind !== FF.NoInteraction && r.kind !== FF.CoulombEwaldDirect]
        (length(real) > 1 || (length(real) == 1 && real[1].kind !== FF.LennardJones)) && return false
    end
    true
end

"GPU replacement of K + 1 loop nests (src/grids.jl:144-150 per probe, :171-177): fills `vgrids[k]` for `probes[k]` and, if given, `cgrid`."
function fill_grids_multi!(vgrids::Vector{Array{Cfloat,4}}, cgrid::Union{Nothing,Array{Cfloat,4}}, probes::Vector{<:ProbeSystem},
                           probe_coulomb::Union{Nothing,ProbeSystem}, ewald::Union{Nothing,EwaldFramework}, cset::GridCoordinatesSetup)
    ref = probes[1]
    ff = ref.forcefield
    tables = map(probes) do p
        check_rules(ff, p.probe, p.atomkinds)
        rule_table(ff, p.probe)
    end
    _, ortho, safemin = CEG.prepare_periodic_distance_computations(ref.mat)
    cutoff2 = NoUnits(ff.cutoff^2/u"Å^2")
    dims, size, shift, Δ = _geometry(cset)
    pos = _flatpos(ref)
    kinds = Int64.(ref.atomkinds)
    mat = Vector{Float64}(vec(ref.mat)); invmat = Vector{Float64}(vec(ref.invmat))
    λv = inv(GRID_TO_KELVIN); thrv = GRID_TO_KELVIN*1e7                                    # src/grids.jl:141-143,148
    λc = ustrip(u"K*Å/e_au^2", COULOMBIC_CONVERSION_FACTOR)/GRID_TO_KELVIN; thrc = inv(λc)*1e7   # :169-170
    α = ewald isa Nothing ? 0.0 : NoUnits(ewald.α*u"Å")
    charges = probe_coulomb isa Nothing ? Float64[] : probe_coulomb.charges
    rules = [t[1] for t in tables]; offsets = [t[2] for t in tables]
    GC.@preserve vgrids cgrid pos kinds charges mat invmat rules offsets dims size shift Δ begin
        rules_pp = Ptr{Cvoid}[pointer(r) for r in rules]
        offs_pp = Ptr{Cvoid}[pointer(o) for o in offsets]
        out_pp = Ptr{Cvoid}[pointer(g) for g in vgrids]
        GC.@preserve rules_pp offs_pp out_pp _check(ccall((:ceg_grids_multi, LIB[]), Cint,
            (Ptr{Float64}, Ptr{Int64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Int32, Float64, Float64,
             Int32, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Int32, Float64,
             Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Float64, Float64, Float64, Float64, Ptr{Ptr{Cvoid}}, Ptr{Cfloat}, Int32),
            pos, kinds, (probe_coulomb isa Nothing ? C_NULL : pointer(charges)), length(kinds), mat, invmat, ortho, safemin^2, cutoff2,
            length(probes), rules_pp, offs_pp, length(offsets[1])-1, α,
            dims, size, shift, Δ,
            λv, thrv, λc, thrc, out_pp, (cgrid isa Nothing ? C_NULL : pointer(cgrid)), ngpus()))
    end
    vgrids, cgrid
end

"""
    create_grids_multi(vdw_files, coulomb_file, framework, forcefield, spacing, atoms, _ewald=nothing)

`create_grid_vdw` (src/grids.jl:137-157) for every atom of `atoms` (1 to 4, of any rule class) and, unless
`coulomb_file === nothing`, `create_grid_coulomb` (:159-185) in one GPU pass; the files are written by the reference's own lines --
each to a temporary name in its directory, renamed onto the target only when ALL are complete (the cache of src/raspa.jl:426 looks no
further than `isfile`: a truncated file at a cache path would be "retrieved" ever after); on failure the temporaries are removed.
"""
function create_grids_multi(vdw_files, coulomb_file, framework::AbstractSystem{3}, forcefield::ForceField, spacing::TÅ, atoms::Vector{Symbol}, _ewald=nothing)
    cset, num_unitcell = CEG._setup_grid_common(framework, spacing, forcefield.cutoff)
    newgrid() = result_array(cset)
    probes = [ProbeSystem(framework, forcefield, atom) for atom in atoms]
    vgrids = [newgrid() for _ in atoms]
    tmps = String[]
    targets = String[]
    try
        if coulomb_file === nothing
            fill_grids_multi!(vgrids, nothing, probes, nothing, nothing, cset)
        else
            _, num_unitcell_c = CEG._setup_grid_common(framework, spacing, 12.0u"Å")       # src/grids.jl:160: 12 Å whatever the force field says
            ewald = _ewald isa EwaldFramework ? _ewald : CEG.initialize_ewald(framework, num_unitcell_c)
            cgrid = newgrid()
            push!(vgrids, cgrid)           # (released with the others below; only the first length(atoms) entries are VdW grids)
            fill_grids_multi!(vgrids[1:length(atoms)], cgrid, probes, ProbeSystem(framework, forcefield), ewald, cset)
            tmp = string(coulomb_file, ".tmp.", getpid(), ".c")
            push!(tmps, tmp); push!(targets, String(coulomb_file))
            open(tmp, "w") do f
                CEG._create_grid_common(f, cset, num_unitcell_c)
                write(f, ewald.precision)
                write(f, cgrid)
                write(f, NoUnits.(cset.cell.mat./u"Å"))
            end
        end
        for (n, (file, grid)) in enumerate(zip(vdw_files, vgrids[1:length(atoms)]))
            tmp = string(file, ".tmp.", getpid(), ".", n)
            push!(tmps, tmp); push!(targets, String(file))
            open(tmp, "w") do f
                CEG._create_grid_common(f, cset, num_unitcell)
                write(f, grid)
                write(f, NoUnits.(cset.cell.mat./u"Å"))
            end
        end
        foreach(((t, f),) -> mv(t, f; force=true), zip(tmps, targets))
    catch
        foreach(t -> rm(t; force=true), tmps)
        rethrow()
    finally
        foreach(release!, vgrids)          # nothing is returned: the page-locked arrays go back to the library at once
    end
    nothing
end

"""
    prebuild_grids!(framework, pff, syst_mol; gridstep=0.15u"Å", supercell=nothing, new=false, cutoff=12.0u"Å")

Call with the arguments of `setup_RASPA` (src/raspa.jl:472-531) right before it: every grid that `setup_RASPA` would have to
create -- same paths (`grid_locations`, :403-419), same conditions (`retrieve_or_create_grid`, :420-439) -- is created here by
`create_grids_multi`, four probes at a time whatever their rule classes; `setup_RASPA` then only retrieves.
"""
function prebuild_grids!(framework, pff, syst_mol; gridstep=0.15u"Å", supercell=nothing, new=false, cutoff=12.0u"Å")
    (framework isa AbstractMatrix || isinf(cutoff) || cutoff != 12.0u"Å") && return nothing
    syst_framework = CEG.load_framework_RASPA(framework, pff)
    supercell = supercell isa Nothing ? CEG.find_supercell(syst_framework, cutoff) : supercell
    forcefield = CEG._ff(pff; cutoff)
    atoms = unique(syst_mol[:,:atomic_symbol])
    coulomb_grid_path, vdws = CEG.grid_locations(framework, pff, forcefield, atoms, gridstep, supercell)
    needcoulomb = any(!iszero(syst_mol[i,:atomic_charge])::Bool for i in 1:length(syst_mol))
    todo = [i for (i, atom) in enumerate(atoms) if CEG.needsvdwgrid(forcefield, atom) && (new || !isfile(vdws[i]))]
    want_c = needcoulomb && (new || !isfile(coulomb_grid_path))
    groups = [todo[lo:min(lo+3, length(todo))] for lo in 1:4:length(todo)]
    for (n, part) in enumerate(groups)
        with_c = want_c && n == 1
        length(part) + with_c < 2 && continue
        foreach(p -> mkpath(dirname(p)), vdws[part]); with_c && mkpath(dirname(coulomb_grid_path))
        create_grids_multi(vdws[part], with_c ? coulomb_grid_path : nothing, syst_framework, forcefield, gridstep, atoms[part],
                           with_c ? CEG.initialize_ewald(syst_framework, supercell) : nothing)
    end
    nothing
end

"Override the package's two grid builders with the GPU versions (method overwrite)."
function install!()
    @eval CEG begin
        create_grid_vdw(file, framework::AbstractSystem{3}, forcefield::ForceField, spacing::TÅ, atom::Symbol) =
            $(create_grid_vdw)(file, framework, forcefield, spacing, atom)
        create_grid_coulomb(file, framework::AbstractSystem{3}, forcefield::ForceField, spacing::TÅ, _ewald=nothing) =
            $(create_grid_coulomb)(file, framework, forcefield, spacing, _ewald)
    end
    nothing
end

# ------------------------------------------------------------------ consumers of the grids (SURVEY 8f rows f1-f3)
# Batched versions of interpolate_grid (src/grids.jl:212-273), compute_ewald(ctx) for one rigid molecule
# (src/ewald.jl:555-577) and single_contribution_vdw (src/energy.jl:397-427).  Handles are opaque pointers.

_pts(positions) = Float64[NoUnits(x/u"Å") for p in positions for x in p]

"`ceg_interp_create` from a parsed `EnergyGrid` (the array is already in K, src/grids.jl:78)."
function interp_handle(g::CEG.EnergyGrid; device=0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    dims, size, shift, _ = _geometry(g.csetup)
    mat = Vector{Float64}(vec(NoUnits.(g.csetup.cell.mat ./ u"Å")))
    invmat = Vector{Float64}(vec(NoUnits.(g.csetup.cell.invmat .* u"Å")))
    GC.@preserve g dims size shift mat invmat _check(ccall((:ceg_interp_create, LIB[]), Cint,
        (Ref{Ptr{Cvoid}}, Int32, Ptr{Cfloat}, Int32, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32),
        h, device, g.grid, 0, dims, size, shift, mat, invmat, g.ewald_precision == Inf))
    h[]
end

# struct ceg_grid_header (include/ceg_hip.h): what parse_grid (src/grids.jl:61-94) reads besides the payload
struct CegGridHeader
    spacing::Float64
    dims::NTuple{3,Int32}
    has_mat::Int32
    size::NTuple{3,Float64}
    shift::NTuple{3,Float64}
    delta::NTuple{3,Float64}
    unitcell::NTuple{3,Float64}
    num_unitcell::NTuple{3,Int32}
    _pad::Int32
    ewald_precision::Float64
    mat::NTuple{9,Float64}
end

"""
    interp_handle_from_file(path, iscoulomb; mat=nothing, device=0) -> (handle, header)

The GPU interpolation handle of a CACHED grid straight from its file ("Retrieved ... grid", src/raspa.jl:426-438): what
`parse_grid(path, iscoulomb, mat)` + `interp_handle` do, without the host array -- the payload is streamed to the device and multiplied
by `GRID_TO_KELVIN` there (src/grids.jl:78).  `mat`: the unit-cell matrix as for `parse_grid` (Å, unitless); `nothing` uses the one
stored at the end of the file.
"""
function interp_handle_from_file(path::AbstractString, iscoulomb::Bool; mat=nothing, device=0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    hdr = Ref{CegGridHeader}()
    m = mat isa Nothing ? Float64[] : Vector{Float64}(vec(Matrix{Float64}(mat)))
    im = mat isa Nothing ? Float64[] : Vector{Float64}(vec(inv(Matrix{Float64}(mat))))
    GC.@preserve m im _check(ccall((:ceg_interp_create_from_file, LIB[]), Cint,
        (Ref{Ptr{Cvoid}}, Int32, Cstring, Int32, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
        h, device, path, iscoulomb, GRID_TO_KELVIN, (mat isa Nothing ? C_NULL : pointer(m)), (mat isa Nothing ? C_NULL : pointer(im)), hdr))
    h[], hdr[]
end

"interpolate_grid(g, p) for every p of `positions` -> Vector{Float64} (K)"
function interp_points(h::Ptr{Cvoid}, positions)
    pts = _pts(positions); out = Vector{Float64}(undef, length(positions))
    GC.@preserve pts out _check(ccall((:ceg_interp_points, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}),
                                      h, pts, length(out), out))
    out
end

"`ceg_recip_create` from an `EwaldFramework` (k-vectors in the order of kspace.kindices, src/ewald.jl:213-236)."
function recip_handle(ef::EwaldFramework; device=0)
    ijk = Int32[]
    for (jy, jz, jxrange, _) in ef.kspace.kindices, jx in jxrange
        push!(ijk, jx, jy, jz)
    end
    sf = ef.StoreRigidChargeFramework
    re, im_ = Vector{Float64}(real.(sf)), Vector{Float64}(imag.(sf))
    ks = Int32[ef.kspace.ks...]; invmat = Vector{Float64}(vec(NoUnits.(ef.invmat .* u"Å")))    # invmat carries Å^-1 (ewald.jl:44)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve ijk re im_ ks invmat _check(ccall((:ceg_recip_create, LIB[]), Cint,
        (Ref{Ptr{Cvoid}}, Int32, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Int32}, Ptr{Float64}),
        h, device, ijk, ef.kfactors, re, im_, length(ef.kfactors), ks, invmat))
    h[]
end

"compute_ewald for `placements` (each a vector of atom positions) of a molecule with `charges`; `enc`, `static` = ctx.energy_net_charges, ctx.static_contribution[] in K"
function recip_energy(h::Ptr{Cvoid}, placements, charges::Vector{Float64}, enc::Float64, static::Float64)
    pts = Float64[NoUnits(x/u"Å") for mol in placements for p in mol for x in p]
    out = Vector{Float64}(undef, length(placements))
    GC.@preserve pts charges out _check(ccall((:ceg_recip_energy, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32, Int64, Float64, Float64, Ptr{Float64}),
        h, pts, charges, length(charges), length(out), enc, static, out))
    out
end

"single_contribution_vdw of a molecule with ff indices `idx2` at every trial placement; `h` from ceg_pairs_create / ceg_pairs_set_atoms"
function pairs_energy(h::Ptr{Cvoid}, placements, idx2::Vector{Int}, exclude_molecule::Int)
    pts = Float64[NoUnits(x/u"Å") for mol in placements for p in mol for x in p]
    kinds = Int32.(idx2 .- 1); out = Vector{Float64}(undef, length(placements))
    GC.@preserve pts kinds out _check(ccall((:ceg_pairs_energy, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int32}, Int32, Int64, Int32, Ptr{Float64}),
        h, pts, kinds, length(kinds), length(out), exclude_molecule, out))
    out
end

# ------------------------------------------------------------------ Monte-Carlo inner loop (BASELINE config 5)
# Device-resident twin of the energy state of a MonteCarloSetup: movement_energy (src/montecarlo.jl:563-579) in one launch,
# update_mc! (src/montecarlo.jl:615-628) on the device.  The driver (src/simulation.jl:727-781) keeps proposing and accepting.

"`ceg_mc_create` + `ceg_mc_set_guests` from a `MonteCarloSetup` after `baseline_energy(mc)`; `vdw`/`coulomb` from `interp_handle`"
function mc_handle(mc::CEG.MonteCarloSetup, vdw::Vector{Ptr{Cvoid}}, coulomb::Ptr{Cvoid}; device=0)
    step = mc.step; ff = step.ff
    nkinds = size(ff.interactions, 1)
    charge = Float64[k <= length(step.charges) ? NoUnits(step.charges[k]/u"e_au") : 0.0 for k in 1:nkinds]
    charge[isnan.(charge)] .= 0.0
    flat = CegRule[]; offsets = Int32[0]
    for a in 1:nkinds, b in 1:nkinds
        for r in _rules(ff.interactions[a, b])
            p = r.params
            push!(flat, CegRule(Int32(Int(r.kind)), 0, get(p, 1, 0.0), get(p, 2, 0.0), get(p, 3, 0.0), r.shift))
        end
        push!(offsets, Int32(length(flat)))
    end
    mat = Vector{Float64}(vec(NoUnits.(step.mat ./ u"Å"))); invmat = Vector{Float64}(vec(inv(NoUnits.(step.mat ./ u"Å"))))
    ef = mc.ewald.ctx.eframework
    ijk = Int32[]
    for (jy, jz, jxrange, _) in ef.kspace.kindices, jx in jxrange
        push!(ijk, jx, jy, jz)
    end
    sf = ef.StoreRigidChargeFramework
    re, im_ = Vector{Float64}(real.(sf)), Vector{Float64}(imag.(sf))
    ks = Int32[ef.kspace.ks...]; einv = Vector{Float64}(vec(NoUnits.(ef.invmat .* u"Å")))
    cutoff2 = NoUnits(ff.cutoff^2/u"Å^2")
    coulombic = ustrip(u"K*Å/e_au^2", COULOMBIC_CONVERSION_FACTOR)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve vdw charge mat invmat flat offsets ijk re im_ ks einv _check(ccall((:ceg_mc_create, LIB[]), Cint,
        (Ref{Ptr{Cvoid}}, Int32, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}, Float64,
         Ptr{CegRule}, Ptr{Int32}, Float64, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Int32}, Ptr{Float64}),
        h, device, vdw, coulomb, charge, nkinds, mat, invmat, cutoff2, flat, offsets, coulombic,
        ijk, ef.kfactors, re, im_, length(ef.kfactors), ks, einv))
    # guests in the order of the Ewald indices ij (mc.revflatidx)
    pos = Float64[]; kinds = Int32[]; first = Int32[0]
    for (i, j) in mc.revflatidx
        for (k, l) in enumerate(step.posidx[i][j])
            append!(pos, NoUnits.(step.positions[l] ./ u"Å")); push!(kinds, Int32(step.ffidx[i][k] - 1))
        end
        push!(first, Int32(length(kinds)))
    end
    GC.@preserve pos kinds first _check(ccall((:ceg_mc_set_guests, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Int32), h[], pos, kinds, first, length(first)-1))
    h[]
end

"movement_energy of species `ij` (Ewald index, 1-based) where it is (column 1) and at `newpos` (column 2); rows: framework vdw, framework direct, inter, reciprocal (K)"
function mc_trial(h::Ptr{Cvoid}, ij::Int, newpos)
    pts = _pts(newpos); out = Matrix{Float64}(undef, 4, 2)
    GC.@preserve pts out _check(ccall((:ceg_mc_trial, LIB[]), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Ptr{Float64}),
                                      h, ij - 1, pts, 1, out))
    out
end

"""
`n` trial placements of species `ij` that already lie in DEVICE memory (`d_trial`: 3 x m x n Float64, e.g. `pointer(::ROCArray)`),
rows into device memory `d_out` (4 x (n + 1) Float64, column 1 = where it is), enqueued on `stream` (a `hipStream_t`, `C_NULL` = null stream):
no copy, no synchronisation.
"""
function mc_trial_device(h::Ptr{Cvoid}, ij::Int, d_trial::Ptr{Float64}, n::Integer, d_out::Ptr{Float64}, stream::Ptr{Cvoid}=C_NULL)
    _check(ccall((:ceg_mc_trial_device, LIB[]), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Cvoid}),
                 h, ij - 1, d_trial, n, d_out, stream))
    nothing
end

"the same for a NEW species with ff indices `idx` (rows 4 x n)"
function mc_trial_insert_device(h::Ptr{Cvoid}, idx::Vector{Int}, d_trial::Ptr{Float64}, n::Integer, d_out::Ptr{Float64}, stream::Ptr{Cvoid}=C_NULL)
    kinds = Int32.(idx .- 1)
    GC.@preserve kinds _check(ccall((:ceg_mc_trial_insert_device, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Int32}, Int32, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Cvoid}), h, kinds, length(kinds), d_trial, n, d_out, stream))
    nothing
end

"update_mc!(mc, idx, newpos) for a displacement, on the device (asynchronous)"
function mc_accept(h::Ptr{Cvoid}, ij::Int, newpos)
    pts = _pts(newpos)
    GC.@preserve pts _check(ccall((:ceg_mc_accept, LIB[]), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), h, ij - 1, pts))
    nothing
end

"movement_energy of a NEW species with ff indices `idx` at `newpos` (ij < 0 in src/ewald.jl:704-728) -> 4 energies"
function mc_trial_insert(h::Ptr{Cvoid}, idx::Vector{Int}, newpos)
    pts = _pts(newpos); kinds = Int32.(idx .- 1); out = Vector{Float64}(undef, 4)
    GC.@preserve pts kinds out _check(ccall((:ceg_mc_trial_insert, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Int32}, Int32, Ptr{Float64}, Int64, Ptr{Float64}), h, kinds, length(kinds), pts, 1, out))
    out
end

"add_one_system! on the device; returns the new Ewald index ij (1-based)"
function mc_insert(h::Ptr{Cvoid}, idx::Vector{Int}, newpos)
    pts = _pts(newpos); kinds = Int32.(idx .- 1); ij = Ref{Int32}(-1)
    GC.@preserve pts kinds _check(ccall((:ceg_mc_insert, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Int32}, Int32, Ptr{Float64}, Ref{Int32}), h, kinds, length(kinds), pts, ij))
    Int(ij[]) + 1
end

"remove_one_system! on the device; returns oldij like src/ewald.jl:404-413 (the species that now answers to `ij`)"
function mc_remove(h::Ptr{Cvoid}, ij::Int)
    moved = Ref{Int32}(-1)
    _check(ccall((:ceg_mc_remove, LIB[]), Cint, (Ptr{Cvoid}, Int32, Ref{Int32}), h, ij - 1, moved))
    Int(moved[]) + 1
end

"(bins per axis, capacity) of the guest neighbour cells (the CellListMap branch of src/energy.jl:399-404), or `nothing` when the MC cell is small enough for the exhaustive loop"
function mc_neighbour_cells(h::Ptr{Cvoid})
    nb = zeros(Int32, 3)
    cap = Ref{Int32}(0)
    rc = ccall((:ceg_mc_neighbour_cells, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ref{Int32}), h, nb, cap)
    rc < 0 && _check(rc)
    rc == 1 ? (Tuple(Int.(nb)), Int(cap[])) : nothing
end


# ------------------------------------------------------------------ blocking masks (SURVEY 8f row f4)
_to_bitarray(mask::Vector{UInt8}, a, b, c) = BitArray(permutedims(reshape(mask, c, b, a), (3, 2, 1)) .!= 0)

"BlockFile(g::EnergyGrid) (src/grids.jl:188-204) with the cell scan on the GPU"
function blockfile(g::CEG.EnergyGrid; device=0)
    a, b, c = g.csetup.dims .+ 1
    value = Array{Cfloat}(@view g.grid[:, :, :, 1])                 # [z, y, x] column-major = C order [x][y][z]
    mask = Vector{UInt8}(undef, a*b*c)
    dims = Int32[g.csetup.dims...]
    GC.@preserve value mask dims _check(ccall((:ceg_block_from_grid, LIB[]), Cint,
        (Int32, Ptr{Cfloat}, Int32, Ptr{Int32}, Float64, Ptr{UInt8}), device, value, 0, dims, 5e6, mask))
    CEG.BlockFile(g.csetup, _to_bitarray(mask, a, b, c))
end

"The scan of parse_blockfile (src/coordinates.jl:139-152): `protoblocks` as built at :123-133"
function block_spheres(csetup::GridCoordinatesSetup, centers::Vector{SVector{3,Float64}}, radius2::Vector{Float64}; device=0)
    a, b, c = csetup.dims .+ 1
    mat = NoUnits.(csetup.cell.mat ./ u"Å"); invmat = NoUnits.(csetup.cell.invmat .* u"Å")
    _, ortho, safemin = CEG.prepare_periodic_distance_computations(mat)
    dims, _, shift, Δ = _geometry(csetup)
    cs = Float64[x for p in centers for x in p]
    m = Vector{Float64}(vec(mat)); im = Vector{Float64}(vec(invmat))
    mask = Vector{UInt8}(undef, a*b*c)
    GC.@preserve dims Δ shift m im cs radius2 mask _check(ccall((:ceg_block_spheres, LIB[]), Cint,
        (Int32, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Float64, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{UInt8}),
        device, dims, Δ, shift, m, im, ortho, safemin^2, cs, radius2, length(radius2), mask))
    CEG.BlockFile(csetup, _to_bitarray(mask, a, b, c))
end

# ------------------------------------------------------------------ build + file in one call (row f4)
"create_grid_vdw with the .grid file written by the library while the grid is built (ceg_grid_vdw_file)"
function create_grid_vdw_streamed(file, framework::AbstractSystem{3}, forcefield::ForceField, spacing::TÅ, atom::Symbol)
    cset, num_unitcell = CEG._setup_grid_common(framework, spacing, forcefield.cutoff)
    grid = Array{Cfloat,4}(undef, cset.dims[3]+1, cset.dims[2]+1, cset.dims[1]+1, 8)
    probe = ProbeSystem(framework, forcefield, atom)
    ff = probe.forcefield
    check_rules(ff, probe.probe, probe.atomkinds)
    rules, offsets = rule_table(ff, probe.probe)
    _, ortho, safemin = CEG.prepare_periodic_distance_computations(probe.mat)
    λ = inv(GRID_TO_KELVIN); thr = GRID_TO_KELVIN*1e7                       # grids.jl:141-143,148 (GRID_TO_KELVIN is a plain Float64, constants.jl:20)
    io = IOBuffer(); CEG._create_grid_common(io, cset, num_unitcell); header = take!(io)   # grids.jl:108-116
    trailer = reinterpret(UInt8, Vector{Float64}(vec(NoUnits.(cset.cell.mat ./ u"Å")))) |> collect   # :154
    dims, size, shift, Δ = _geometry(cset)
    pos = _flatpos(probe); kinds = Int64.(probe.atomkinds)
    mat = Vector{Float64}(vec(probe.mat)); invmat = Vector{Float64}(vec(probe.invmat))
    GC.@preserve grid pos kinds mat invmat rules offsets dims size shift Δ header trailer begin
        _check(ccall((:ceg_grid_vdw_file, LIB[]), Cint,
            (Ptr{Float64}, Ptr{Int64}, Int64, Ptr{Float64}, Ptr{Float64}, Int32, Float64, Float64,
             Ptr{CegRule}, Ptr{Int32}, Int32, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Float64, Float64, Ptr{Cfloat}, Int32, Cstring, Ptr{UInt8}, Int64, Ptr{UInt8}, Int64),
            pos, kinds, length(kinds), mat, invmat, ortho, safemin^2, NoUnits(ff.cutoff^2/u"Å^2"),
            rules, offsets, length(offsets)-1, dims, size, shift, Δ, λ, thr, grid, ngpus(),
            String(file), header, length(header), trailer, length(trailer)))
    end
    grid
end

end # module
