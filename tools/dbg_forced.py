"""Diagnostic: per-state detail of tests/test_gpu_parity.py::test_forced_contacts_one_substep."""
import importlib.util, json, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import BLOB
from flybody_amd.batched_env import BatchedFlyEnv
from flybody_amd.tasks.synthetic import base_wing_pattern
from flybody_amd.tasks.wbpg import build_tables
from oracle import oracle as O
spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
names = json.load(open(BLOB.replace(".ffmb", ".json")))["geom_name"]
tables = build_tables(base_wing_pattern())
om, ref, kinds = m._forced_contact_states(tables)
dd = O.OracleData(om)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for kind, states in kinds.items():
    B = len(states)
    env = BatchedFlyEnv(tables, *ref, batch_size=B, seed=3)
    env.reset()
    env.set_state(torch.tensor(np.stack([s[0] for s in states])), torch.tensor(np.stack([s[1] for s in states])))
    env.physics_step(torch.tensor(np.stack([s[2] for s in states]).astype(np.float32), device="cuda"), NS)
    q, v = [x.cpu().numpy() for x in env.get_state()]
    ints = env.get_task_state()[0].cpu().numpy()
    print("==", kind)
    cfbuf = None
    if "dbgcf" in os.environ.get("FLYBODY_ENV_LIB", ""):
        import ctypes as C
        from flybody_amd import _capi
        cfbuf = np.zeros((64, 4 + 72), np.float32)
        _capi.lib().ffe_debug_read_cf(cfbuf.ctypes.data_as(C.POINTER(C.c_float)))
    for i, s in enumerate(states):
        dd.qpos[:], dd.qvel[:], dd.ctrl[:] = s
        dd.step1()
        con = [(names[int(c[0])][:-10], names[int(c[1])][:-10], f"{c[5]:.2e}", int(c[3])) for c in dd.contacts()]
        J, aref, D, ty = dd.efc()
        dd.step2()
        c0 = dd.contacts()
        frc = np.ctypeslib.as_array(om.L.fo_efc_force(dd.ptr), (dd.nefc,)).copy()
        if cfbuf is not None:
            o = cfbuf[i]
            print(f"  gpu: nct {int(o[0])} active mask {int(o[1]):#b} iters {int(o[2])} active limits {int(o[3])}; oracle nefc {dd.nefc} forces {np.round(frc, 2).tolist()}")
            for k in range(int(o[0])):
                q = o[4 + 12 * k: 16 + 12 * k]
                pid = int(q[7])
                print(f"    gpu {names[pid & 255][:-10]}|{names[pid >> 8][:-10]} dist {q[0]:.3e} n {np.round(q[1:4], 4).tolist()} pos {np.round(q[4:7], 4).tolist()} D {q[8]:.4e} aref {q[9]:.4e} force {q[10]:.4f} incl {q[11]:.2e}")
            for cr in c0:
                print(f"    ora {names[int(cr[0])][:-10]}|{names[int(cr[1])][:-10]} dist {cr[5]:.3e} n {np.round(cr[9:12], 4).tolist()} pos {np.round(cr[6:9], 4).tolist()} excl {int(cr[3])} force {cr[15]:.4f}")
        dd.step1()
        for _ in range(NS - 1):
            dd.step2(); dd.step1()
        ev = np.abs(v[i] - dd.qvel)
        print(i, f"qvel err {ev.max():.2e} (rel {(ev / np.maximum(1, np.abs(dd.qvel))).max():.1e}) at {ev.argsort()[-3:].tolist()} | gpu word {ints[i,7]:#x} iters {ints[i,6]} | oracle rows types {ty.tolist()} forces {np.round(frc, 2).tolist()} | contacts {con}")
    env.close()
