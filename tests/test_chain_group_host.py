"""Host side of the chain groups (pymc_amd/chain_group.py) that needs no device: what is not even offered to the engine."""

from pymc_amd.chain_group import ChainGroup


def test_a_group_is_two_to_four_chains():
    # (`try_create` answers None -- "run them as independent engines" -- without building anything)
    assert ChainGroup.try_create([]) is None
    assert ChainGroup.try_create([object()]) is None
    assert ChainGroup.try_create([object()] * 5) is None
    assert ChainGroup.MAX_CHAINS == 4


def test_group_symbols_are_part_of_the_bound_abi():
    from pymc_amd import _lib

    for name in ("nuts_group_create", "nuts_group_add", "nuts_group_remove", "nuts_group_destroy", "nuts_group_launches"):
        assert name in _lib.SYMBOLS
