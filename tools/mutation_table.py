"""profiles/r05_mutation_table.txt: which checks of which fixture family notice which injected numeric fault
(tests/mutations.py; asserted by tests/test_oracle_cpu.py::test_parity_fixtures_detect_injected_numeric_faults).

    python tools/mutation_table.py > profiles/r05_mutation_table.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.mutations import MUTATIONS, detect  # noqa: E402
from tests.test_oracle_cpu import MUTATION_CASES  # noqa: E402


def main():
    print("Mutation self-check of the Dual-AR parity fixtures (CPU oracle with an injected fault vs the committed fixture).")
    print("tokens = entries of the free-running token matrix that differ from the fixture (first differing frame);")
    print("taps   = the teacher-forced float-tap check of tests/helpers.check_teacher_forced (16 bf16 steps, rel. L2 2 %,")
    print("         near-argmax / equal-draw decisions) FAILS, with its first complaint.")
    print()
    print(f"{'fixture':16s} {'fault':22s} {'tokens':>14s}  taps")
    for case in MUTATION_CASES:
        for m in (None,) + MUTATIONS:
            r = detect(case, m)
            tok = "0" if r["tokens_changed"] == 0 else f"{r['tokens_changed']} (f{r['first_token_mismatch_frame']})"
            taps = "n/a" if r["taps_fail"] is None else ("FAIL: " + r["taps_reason"] if r["taps_fail"] else "pass")
            print(f"{case:16s} {str(m or 'none (clean)'):22s} {tok:>14s}  {taps}")
        print()
    print("S2-width fixtures (dualar_s2_*.npz) carry tokens only; the float check at that width is")
    print("tests/test_dualar_gpu.py::test_s2_shape_random_weights_prompt_of_250_and_decode_positions_across_a_page_boundary")
    print("(random weights, calibrated against the fp32-exact oracle), which also injects a fault into the HIP model and")
    print("requires the criterion to fail.")


if __name__ == "__main__":
    main()
