"""CPU tests of the drop-in boundary: libmtg_hip.so loads, exports every symbol
include/mtg.h declares, and fails loudly (never falls back) without a GPU."""
import os
import re

import pytest

from mind_the_gaps_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mtg.h")).read()
    return sorted(set(re.findall(r"MTG_API\s+[\w\s\*]+?\b(mtg_\w+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = engine.load_library()
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "include/mtg.h declares %s but libmtg_hip.so does not export it" % n
    assert sorted(engine.EXPORTS) == names


def test_constants_match_header():
    text = open(os.path.join(ROOT, "include", "mtg.h")).read()
    defs = dict(re.findall(r"#define\s+(MTG_\w+)\s+\(?(-?\d+)\)?", text))
    for name, value in (("MTG_TERM_REAL", engine.TERM_REAL), ("MTG_TERM_SHO", engine.TERM_SHO),
                        ("MTG_TERM_DRW", engine.TERM_DRW), ("MTG_TERM_LORENTZIAN", engine.TERM_LORENTZIAN),
                        ("MTG_TERM_COSINUS", engine.TERM_COSINUS), ("MTG_TERM_BPL", engine.TERM_BPL),
                        ("MTG_TERM_MATERN32", engine.TERM_MATERN32), ("MTG_TERM_JITTER", engine.TERM_JITTER),
                        ("MTG_MEAN_LINEAR", engine.MEAN_LINEAR), ("MTG_ST_NOTPD", engine.ST_NOTPD),
                        ("MTG_ST_PRIOR", engine.ST_PRIOR), ("MTG_E_UNSUPPORTED", engine.E_UNSUPPORTED)):
        assert int(defs[name]) == value, name


def test_pure_queries_need_no_gpu():
    lib = engine.load_library()
    assert lib.mtg_version().startswith(b"mtg-hip")
    assert [lib.mtg_term_nparams(k) for k in range(10)] == [2, 3, 4, 3, 2, 1, 2, 3, 2, 3]
    assert lib.mtg_term_nparams(99) == -1
    # compiled structure table: J = jr + 2 jc <= 10
    assert lib.mtg_structure_supported(1, 2) == 1
    assert lib.mtg_structure_supported(0, 5) == 1
    assert lib.mtg_structure_supported(10, 0) == 1
    assert lib.mtg_structure_supported(0, 0) == 1      # a white kernel (JitterTerm alone): mtg_white_kernel
    assert lib.mtg_structure_supported(1, 5) == 0


@pytest.mark.skipif(engine.device_count() > 0, reason="a GPU is present")
def test_fails_loudly_without_gpu():
    with pytest.raises(engine.EngineUnavailable):
        engine.Engine(0)


def test_sharding_entry_points_check_their_arguments_without_a_gpu():
    """mtg_ensemble_shard_* / mtg_rccl_unique_id with no context or no buffer: MTG_E_ARG, nothing touched."""
    import ctypes
    lib = engine.load_library()
    noop = engine.EXCHANGE_FN(lambda user, lnp, status, count, lo, hi: 0)
    assert lib.mtg_ensemble_shard_host(None, 0, 1, noop, None) == engine.E_ARG
    assert lib.mtg_ensemble_shard_rccl(None, ctypes.create_string_buffer(128), 0, 1) == engine.E_ARG
    assert lib.mtg_ensemble_unshard(None) == engine.E_ARG
    assert lib.mtg_rccl_unique_id(None) == engine.E_ARG


def test_rocfft_seed_cache_is_copied_per_process(monkeypatch, tmp_path):
    """rocFFT writes to the cache file it is given: the tracked seed must never be that file.  The loader hands
    rocFFT a private copy in the temporary directory -- if the seed was made with the librocfft this process will
    load (version stamp next to it) -- and leaves a path chosen by the user alone."""
    import os
    from mind_the_gaps_amd import engine
    monkeypatch.delenv("ROCFFT_RTC_CACHE_PATH", raising=False)
    monkeypatch.setattr(engine, "rocfft_cache_seeded", False)
    engine._seed_rocfft_cache()
    stamp_ok = os.path.exists(engine.ROCFFT_CACHE_STAMP) and \
        open(engine.ROCFFT_CACHE_STAMP).read().strip() == engine.rocfft_library_version()
    if stamp_ok:
        copy = os.environ["ROCFFT_RTC_CACHE_PATH"]
        assert engine.rocfft_cache_seeded and os.path.abspath(copy) != os.path.abspath(engine.ROCFFT_CACHE_SEED)
        assert os.path.getsize(copy) == os.path.getsize(engine.ROCFFT_CACHE_SEED) and str(os.getpid()) in os.path.basename(copy)
        os.remove(copy)
    else:
        assert not engine.rocfft_cache_seeded and "ROCFFT_RTC_CACHE_PATH" not in os.environ
    # a seed made with another rocFFT build is skipped
    monkeypatch.delenv("ROCFFT_RTC_CACHE_PATH", raising=False)
    monkeypatch.setattr(engine, "ROCFFT_CACHE_STAMP", str(tmp_path / "other.version"))
    (tmp_path / "other.version").write_text("librocfft.so.0.0.0:1\n")
    engine._seed_rocfft_cache()
    assert not engine.rocfft_cache_seeded and "ROCFFT_RTC_CACHE_PATH" not in os.environ
    # the user's own choice wins
    monkeypatch.setenv("ROCFFT_RTC_CACHE_PATH", str(tmp_path / "mine.db"))
    engine._seed_rocfft_cache()
    assert os.environ["ROCFFT_RTC_CACHE_PATH"] == str(tmp_path / "mine.db") and not engine.rocfft_cache_seeded
