import sys, numpy as np
sys.path.insert(0, "webgl-fluid-simulation_amd"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import fluid_hip, scenario as S
from oracle import oracle as O
def half(a):
    with np.errstate(over="ignore"): return np.asarray(a,np.float32).astype(np.float16).astype(np.float32)
# flip fractions per pass
W,H=512,300
rng=np.random.default_rng(5)
st={"velocity": half(rng.normal(0,80,(H,W,2))), "pressure": half(rng.normal(0,30,(H,W))), "divergence": half(rng.normal(0,30,(H,W))),
    "curl": half(rng.normal(0,30,(H,W))), "dye": half(np.abs(rng.normal(0,1,(H,W,4))))}
dt=np.float32(0.016666)
with fluid_hip.FluidSim(canvas=(W,H), config={"SIM_RESOLUTION":300,"DYE_RESOLUTION":300}, schedule="passes", storage="f16") as sim:
    P=sim.params()
    def load(s):
        for k,v in s.items(): sim.write(k,v)
    load(st); sim.run_pass("vorticity"); g=sim.read("velocity"); w=O.round_half(O.vorticity(st["velocity"],st["curl"],P.curl,dt)); print("vorticity flips", (g!=w).mean())
    sm=dict(st, velocity=half(st["velocity"]*0.05))
    load(sm); sim.run_pass("advect_velocity"); g=sim.read("velocity"); w=O.round_half(O.advect(sm["velocity"],sm["velocity"],dt,P.velocity_dissipation)); print("advect vel flips",(g!=w).mean())
    load(sm); sim.run_pass("advect_dye"); g=sim.read("dye"); w=O.round_half(O.advect(sm["velocity"],sm["dye"],dt,P.density_dissipation)); print("advect dye flips",(g!=w).mean())
for curl in (0,30):
  for canvas,cfg in (((256,256),{"SIM_RESOLUTION":64,"DYE_RESOLUTION":64}), ((600,300),{"SIM_RESOLUTION":48,"DYE_RESOLUTION":160}), ((1024,1024),{"SIM_RESOLUTION":256,"DYE_RESOLUTION":256})):
    cfg=dict(cfg,CURL=curl,PRESSURE_ITERATIONS=20)
    ref=O.RefSim(canvas=canvas,config=cfg,seed=11,storage="f16")
    with fluid_hip.FluidSim(canvas=canvas,config=cfg,storage="f16",random=fluid_hip.mulberry32(11)) as sim:
        ref.multiple_splats(4); sim.multipleSplats(4)
        print("splat flips", [(sim.read(k)!=ref.fields()[k]).mean() for k in ("velocity","dye")])
        ref.step(0.016666,3); sim.step(0.016666,3)
        got=sim.fields()
    print("curl",curl,canvas,{k: float("%.2e"%S.rel_err(got[k],ref.fields()[k])) for k in S.FIELDS})
