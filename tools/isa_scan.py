"""Scan device assembly (hipcc -S --cuda-device-only) for signs of loops the kernels expect to be unrolled:
dynamic register indexing (s_set_gpr_idx_on / v_movrel), scratch use; prints per kernel the MFMA count, the other
vector instructions (packed ones apart), cross-lane traffic (v_readlane / v_writelane, ds_bpermute / ds_swizzle, DPP)
and -- with --split -- the same counts before the first / between the first and last / behind the last MFMA (prologue,
body, tail: an opcode total alone says nothing about WHERE, e.g. k_transient_bins' 1 151 readlanes are all tail).
"inflight": the largest number of global loads outstanding along the instruction stream (counted up at every
global_load, cut back to N at every s_waitcnt vmcnt(N); layout order, so an upper-bound sketch) -- a gather kernel whose
lookups were meant to be in flight together and shows 8-11 here has a wait between them (round 3: level records read
through a lane-dependent index became per-lane loads from the argument segment, each behind an s_waitcnt vmcnt(0)).
"canon": v_max_f32 x, x, x -- the canonicalisation the compiler puts in front of fmaxf() on an MFMA result.
usage: python tools/isa_scan.py [--split] file.s [...]"""
import re, sys

args = [a for a in sys.argv[1:] if a != "--split"]
split = "--split" in sys.argv[1:]


def counts(lines):
    ops = [l.split()[0] for l in lines]
    n = lambda pred: sum(1 for o in ops if pred(o))
    return dict(mfma=n(lambda o: o.startswith("v_mfma")), valu=n(lambda o: o.startswith("v_") and not o.startswith("v_mfma")),
                pk=n(lambda o: o.startswith("v_pk_")), readlane=n(lambda o: o.startswith(("v_readlane", "v_writelane", "v_readfirstlane"))),
                bperm=n(lambda o: o.startswith(("ds_bpermute", "ds_permute", "ds_swizzle"))),
                dpp=sum(1 for l in lines if "_dpp" in l.split()[0] or " quad_perm:" in l or " row_" in l or " wave_" in l),
                lds=n(lambda o: o.startswith("ds_")), vmem=n(lambda o: o.startswith(("global_", "buffer_", "flat_"))),
                salu=n(lambda o: o.startswith("s_")), inflight=inflight(lines),
                canon=sum(1 for l in lines if re.match(r'v_max_f32_e32 v\d+, (v\d+), \1$', l)))


def inflight(lines):
    cur = best = 0
    for l in lines:
        if l.startswith("global_load") and "_lds_" not in l.split()[0]:
            cur += 1
            best = max(best, cur)
        else:
            m = re.match(r's_waitcnt .*vmcnt\((\d+)\)', l)
            if m:
                cur = min(cur, int(m.group(1)))
    return best


for f in args:
    s = open(f).read()
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
        body = m.group(2)
        n = body.count('s_set_gpr_idx_on') + body.count('v_movrel')
        if not (n or 'v_mfma' in body or 'scratch_' in body):
            continue
        lines = [l.strip() for l in body.splitlines()]
        lines = [l for l in lines if l and not l.startswith(('.', ';', '//')) and not l.endswith(':')]
        c = counts(lines)
        print(f"{f.split('/')[-1]:16s} {m.group(1)[-56:]:56s} gpr_idx {n:5d}  mfma {c['mfma']:5d}  scratch {len(re.findall(r'scratch_', body)):4d}  "
              f"branches {len(re.findall(r's_cbranch', body)):4d}  valu {c['valu']:6d} (pk {c['pk']:5d})  readlane {c['readlane']:5d}  "
              f"bpermute {c['bperm']:5d}  dpp {c['dpp']:5d}  lds {c['lds']:5d}  vmem {c['vmem']:5d}  inflight {c['inflight']:3d}  canon {c['canon']:4d}")
        if split and c['mfma']:
            idx = [i for i, l in enumerate(lines) if l.startswith('v_mfma')]
            for name, part in (("before the first MFMA", lines[:idx[0]]), ("first..last MFMA", lines[idx[0]:idx[-1] + 1]), ("behind the last MFMA", lines[idx[-1] + 1:])):
                p = counts(part)
                print(f"{'':16s}   {name:24s} mfma {p['mfma']:5d}  valu {p['valu']:6d} (pk {p['pk']:5d})  readlane {p['readlane']:5d}  bpermute {p['bperm']:5d}  "
                      f"dpp {p['dpp']:5d}  lds {p['lds']:5d}  vmem {p['vmem']:5d}  salu {p['salu']:5d}")
