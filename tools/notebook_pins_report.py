"""Table behind tests/test_reference_notebook_pins.py: what the reference's notebooks printed against the exact integrals
of the oracle's model.  python tools/notebook_pins_report.py > profiles/r02/notebook_pins.md"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import tests.test_reference_notebook_pins as T  # noqa: E402

WHERE = {"A": "gpax_simpleGP.ipynb cell 14 (RBF)", "B": "gpax_simpleGP.ipynb cell 27 (RBF)", "C": "gpax_simpleGP.ipynb cell 42 (RBF, Gamma(2,5) length prior)",
         "D": "gpax_UIGP.ipynb cell 12 (Matern, Gamma / HalfNormal priors)", "E": "MeasuredNoiseGP.ipynb cell 11 (Matern, measured variances)",
         "F": "gpax_GPBO.ipynb cell 22, step 1 (RBF, HalfNormal(0.01) noise prior)", "G": "GP_sGP.ipynb cell 18 (Matern)",
         "P3": "simpleGP.ipynb cell 26, period 0.3 (Periodic)", "P6": "simpleGP.ipynb cell 26, period 0.6 (Periodic)",
         "P10": "simpleGP.ipynb cell 26, period 1.0 (Periodic)"}
print("# Outputs the reference's notebooks hold against exact integration of the oracle's model\n")
print("`tests/test_reference_notebook_pins.py` (CPU) / `tests/test_gpu_reference_notebook.py` (the product on the GPU).  "
      "Printed = numpyro `print_summary` of the reference's NUTS run (two decimals); exact = tensor-grid integral of the "
      "posterior of `oracle/cpu_ref.py`'s model; tolerance = 0.005 + 4 x Monte-Carlo error implied by the printed n_eff.\n")
print("| problem | parameter | printed mean / std / median (n_eff) | exact mean / std / median | tolerance on the mean |")
print("|---|---|---|---|---|")
for case, table in T.PRINTED.items():
    ex = T.posterior_marginals(case)
    for name, (mean, std, med, neff) in table.items():
        q = ex[name]
        print(f"| {case}: {WHERE[case]} | {name} | {mean} / {std} / {med} ({neff:.0f}) | {q[0]:.4f} / {q[1]:.4f} / {q[2]:.4f} | "
              f"{0.005 + 4 * q[1] / np.sqrt(neff):.4f} |")
p = T.PRINTED_SVI
at = T._neg_log_joint_A([p["k_length"], p["k_scale"], p["noise"]])
print(f"\nviGP (compare_GPs.ipynb cell 20): printed state after 1000 SVI steps k_length {p['k_length']}, k_scale {p['k_scale']}, "
      f"noise {p['noise']}, average loss of steps 951-1000 {p['avg_loss_951_1000']}; the oracle's negative log joint at that state: "
      f"{at:.4f}.")
