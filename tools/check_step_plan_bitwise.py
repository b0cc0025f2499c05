"""the single-submission step plan against the launch-by-launch step, BITWISE, on the lane emulator with one thread (float atomics in a fixed order):
two ordinary steps, snapshot, two plan steps -- then two ordinary steps from the snapshot.   HIPEMU_THREADS=1 python tools/check_step_plan_bitwise.py
(about ten minutes; the ordinary CPU test compares one step with tolerances under the multi-threaded emulator)"""
import copy, os, sys
os.environ.setdefault("HIPEMU_THREADS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import SEED
from emu_patch import product_on_emulator
from eeg_image_decode_amd import synthetic as syn
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
B, NC, NSTEP = 64, 40, 2
cls = T(syn.unit_features(SEED + 4, NC, tag="c"))
rng = np.random.default_rng(1)
data = [(T(syn.eeg_batch(SEED + 100 + i, B)), T(syn.unit_features(SEED + 200 + i, B, tag="i")), T(syn.unit_features(SEED + 300 + i, B, tag="t")),
         T(rng.integers(0, NC, size=B).astype(np.int64))) for i in range(2 + NSTEP)]
with product_on_emulator():
    from eeg_image_decode_amd import optim, retrieval, step_plan
    from eeg_image_decode_amd.atms import ATMS
    step_plan._runtime_ok = lambda: True
    step_plan._on_device = lambda t: True
    step_plan.StepPlan.WARM_STEPS = 2
    torch.manual_seed(5)
    m = ATMS().train()
    opt = optim.AdamW(m.parameters(), lr=3e-4)
    acc, correct = [], torch.zeros(1, dtype=torch.int32)
    for x, img, txt, lab in data[:2]:
        retrieval.contrastive_step(m, opt, x, 1, img, txt, lab, cls, acc, correct)
    snap = (copy.deepcopy(m.state_dict()), copy.deepcopy(opt.state_dict()), torch.get_rng_state(), correct.clone())
    for x, img, txt, lab in data[2:]:
        retrieval.contrastive_step(m, opt, x, 1, img, txt, lab, cls, acc, correct)
    assert retrieval.step_plans_of(m), "the plan did not engage"
    res = ({k: v.clone() for k, v in m.state_dict().items()}, [float(a) for a in acc[2:]], int(correct))
    os.environ["EEGCLIP_STEP_PLAN"] = "0"
    m2 = ATMS().train()
    m2.load_state_dict(snap[0])
    opt2 = optim.AdamW(m2.parameters(), lr=3e-4)
    opt2.load_state_dict(snap[1])
    torch.set_rng_state(snap[2])
    acc2, correct2 = [], snap[3].clone()
    for x, img, txt, lab in data[2:]:
        retrieval.contrastive_step(m2, opt2, x, 1, img, txt, lab, cls, acc2, correct2)
bad = [k for k, v in m2.state_dict().items() if not torch.equal(v, res[0][k])]
print("losses plan", res[1], "ordinary", [float(a) for a in acc2], "correct", res[2], int(correct2))
print("BIT-IDENTICAL" if not bad and res[1] == [float(a) for a in acc2] and res[2] == int(correct2) else f"DIFFERENT in {bad[:8]} ({len(bad)} tensors)")
