import sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle import orc
from tests import cases, helpers as H
from tests.test_restart_gpu import new_model, step
from mom6_amd import abi
cfg = H.double_gyre(); gg, d, M = cfg
inp = cases.rk2_inputs(cfg, False, False); dt = inp["dt"]; bt_mod = dict(strong_drag=1)
so, m = cases.oracle_rk2(orc, cfg, inp, 2, bt_mod)
dyc, forcing = new_model(cfg, inp, bt_mod)
sg = dict(u=dyc.to_dev(inp["u"]), v=dyc.to_dev(inp["v"]), h=dyc.to_dev(inp["h"]), uh=dyc.zeros3(), vh=dyc.zeros3(), uhtr=dyc.zeros3(), vhtr=dyc.zeros3(), eta_av=dyc.zeros2())
dyc.dyn_split_RK2_new_run(sg["u"], sg["v"], sg["h"], sg["uh"], sg["vh"], dt)
for n in range(2):
    step(dyc, sg, forcing, dt, n == 0)
dyc.sync()
for k in ("u", "v", "h"):
    print("after 2 steps dev vs orc", k, np.abs(sg[k].cpu().numpy() - so[k]).max())
keep = {k: sg[k].clone() for k in ("u", "v", "h", "uhtr", "vhtr")}
names = ("eta", "u_av", "v_av", "diffu", "diffv", "CAu_pred", "CAv_pred")
keep_cs = {k: dyc.rk2_field(k).clone() for k in names}
for k in names:
    print("CS dev vs orc", k, np.abs(keep_cs[k].cpu().numpy() - m[k]).max())
keep_bt = {k: dyc.barotropic_field(k).clone() for k in ("ubtav", "vbtav")}
for k in keep_bt:
    print("BT dev vs orc", k, np.abs(keep_bt[k].cpu().numpy() - m.btcs[k]).max())
dtbt = dyc.barotropic_dtbt(); print("dtbt", dtbt, m.bt.dtbt)
torch.cuda.synchronize(); dyc.close()
for have in (7, 15):
    cont, bt, cor, pgf, rk2 = cases.rk2_params(d, inp["GV"], bt_mod, None, None)
    m2 = orc.OrcModel(d, M, inp["GV"], cont, bt, cor, pgf, rk2, inp["Rlay"], inp["gp"], 0)
    for n in names: m2[n][...] = m[n]
    for n in ("ubtav", "vbtav"): m2.btcs[n][...] = m.btcs[n]
    bt.dtbt = m.bt.dtbt
    so2 = {k: v.copy() for k, v in so.items()}; so2["uh"][...] = 0; so2["vh"][...] = 0
    m2.restart_fills(so2["u"], so2["v"], so2["h"], so2["uh"], so2["vh"], dt, have)
    dyc, forcing = new_model(cfg, inp, bt_mod); dyc.sync()
    sg = dict(u=keep["u"].clone(), v=keep["v"].clone(), h=keep["h"].clone(), uh=dyc.zeros3(), vh=dyc.zeros3(), uhtr=keep["uhtr"].clone(), vhtr=keep["vhtr"].clone(), eta_av=dyc.zeros2())
    for k, a in keep_cs.items(): dyc.rk2_field(k).copy_(a)
    for k, a in keep_bt.items(): dyc.barotropic_field(k).copy_(a)
    dyc.barotropic_dtbt(dtbt)
    torch.cuda.synchronize()
    dyc.dyn_split_RK2_restart_fills(sg["u"], sg["v"], sg["h"], sg["uh"], sg["vh"], dt, have)
    dyc.sync()
    for k in ("h_av", "CAu_pred", "u_av", "eta", "diffu"):
        print(have, "fills dev vs orc", k, np.abs(dyc.rk2_field(k).cpu().numpy() - m2[k]).max())
    for k in ("uh", "vh"):
        print(have, "fills dev vs orc", k, np.abs(sg[k].cpu().numpy() - so2[k]).max())
    for n in range(2):
        step(dyc, sg, forcing, dt, False)
        m2.step(so2["u"], so2["v"], so2["h"], so2["uh"], so2["vh"], so2["uhtr"], so2["vhtr"], so2["eta_av"], inp["taux"], inp["tauy"], dt, inp["coefs"], calc_dtbt=False)
        dyc.sync()
        for k in ("u", "v", "h", "uh", "eta_av"):
            print(have, "step", n, "dev vs orc", k, np.abs(sg[k].cpu().numpy() - so2[k]).max())
        for k in ("PFu", "CAu", "u_accel_bt", "visc_rem_u", "eta", "uhbt", "u_av", "h_av", "eta_PF"):
            print(have, "step", n, "  CS", k, np.abs(dyc.rk2_field(k).cpu().numpy() - m2[k]).max())
    torch.cuda.synchronize(); dyc.close()
