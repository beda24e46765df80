"""CPU tier: the generated median selection network (panorama-opticalflow_amd/tools/gen_median_net.py -> csrc/median_net.inl) is
re-verified -- exhaustively with the 0-1 principle (2^30 inputs) and on random floats with ties -- and the committed .inl is what the
generator produces (medianBlur 5, CPU/PixFlow.hpp:325,338)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_median_network_is_exact_and_current():
    path = os.path.join(ROOT, "panorama-opticalflow_amd", "tools", "gen_median_net.py")
    spec = importlib.util.spec_from_file_location("gen_median_net", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    text, inl = mod.main(write=False, log=lambda *a: None)   # raises if either verification fails
    assert open(inl).read() == text, "csrc/median_net.inl is stale: run tools/gen_median_net.py"
