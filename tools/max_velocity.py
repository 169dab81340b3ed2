import sys, numpy as np
sys.path.insert(0, "webgl-fluid-simulation_amd")
import fluid_hip
size=4096
cfg={"SIM_RESOLUTION": size, "DYE_RESOLUTION": size, "PRESSURE_ITERATIONS": 50}
for canvas in ((size,size),(size, size*8)):
    c=dict(cfg, SIM_RESOLUTION=min(canvas), DYE_RESOLUTION=min(canvas))
    with fluid_hip.FluidSim(canvas=canvas, config=c, random=fluid_hip.mulberry32(1234)) as sim:
        sim.multipleSplats(20)
        mx=[]
        for k in range(6):
            v=sim.read("velocity"); mx.append(float(np.abs(v).max())); del v
            sim.step(0.016666, 50)
        v=sim.read("velocity"); mx.append(float(np.abs(v).max()))
        print(canvas, "max|v| every 50 steps:", [round(m,1) for m in mx], " -> back-trace texels", round(max(mx)*0.016666,2))
