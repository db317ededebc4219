"""Host-side logic of the product library that runs without a GPU."""
import pytest


def test_recipe_batch_range_matches_oracle(capi, oracle):
    """aasr_recipe_batch_range (C++ restatement of Recipe::read's batching,
    aku/Recipe.cc:63-112) against the oracle's Python restatement."""
    for L in range(0, 30):
        text = "\n".join("audio=a%d.wav lna=a%d.lna" % (i, i) for i in range(L)) + "\n"
        for n in range(0, 10):
            for b in range(1, max(n, 1) + 1):
                infos = oracle.recipe_read(text, n, b)
                first, cnt = capi.recipe_batch_range(L, n, b)
                assert cnt == len(infos), (L, n, b)
                if cnt:
                    assert infos[0].audio_path == "a%d.wav" % first


def test_recipe_batch_index_validation(capi):
    with pytest.raises(capi.AasrError, match="Invalid batch index"):
        capi.recipe_batch_range(10, 4, 5)
    with pytest.raises(capi.AasrError, match="Invalid batch index"):
        capi.recipe_batch_range(10, 4, 0)
