"""Host logic of the neighbour halo transport (sbmc_amd/halo.py) that needs no GPU: how a tensor's rows become a
2-d run of bytes and how a run larger than a mailbox slot is cut into messages (both ends of a link must cut
alike: the function is deterministic in the run's shape)."""
import pytest
import torch as th


def test_rows_run_planar_and_channels_last():
    from sbmc_amd.halo import rows_run
    t = th.zeros(2, 3, 10, 8)                                   # planar: one chunk per (image, channel) plane
    ptr, chunks, nbytes, pitch = rows_run(t, 2, 5)
    assert (ptr - t.data_ptr(), chunks, nbytes, pitch) == (2 * 8 * 4, 6, 3 * 8 * 4, 10 * 8 * 4)
    u = th.zeros(2, 4, 10, 8).contiguous(memory_format=th.channels_last)
    ptr, chunks, nbytes, pitch = rows_run(u, 7, 10, nhwc=True)    # channels-last: rows of an image are one block
    assert (ptr - u.data_ptr(), chunks, nbytes, pitch) == (7 * 8 * 4 * 4, 2, 3 * 8 * 4 * 4, 10 * 8 * 4 * 4)
    h = th.zeros(1, 2, 5, 6, 7, dtype=th.float16)               # any leading dimensions, any element size
    assert rows_run(h, 0, 6)[1:] == (10, 6 * 7 * 2, 6 * 7 * 2)
    with pytest.raises(ValueError):
        rows_run(t[..., ::2], 0, 1)                             # not dense
    with pytest.raises(ValueError):
        rows_run(t, 0, 1, nhwc=True)                            # planar strides are not channels-last ones


@pytest.mark.parametrize("chunks,chunk_bytes,slot", [(6, 96, 4096), (14, 288, 4096), (1, 5120, 4096), (3, 10000, 4096),
                                                     (100, 64, 640), (7, 4096, 4096)])
def test_messages_cover_a_run_exactly_once(chunks, chunk_bytes, slot):
    from sbmc_amd.halo import _pieces
    seen = [bytearray(chunk_bytes) for _ in range(chunks)]
    for c0, n, b0, nb in _pieces(chunks, chunk_bytes, slot):
        assert n * nb <= slot and n >= 1 and nb >= 1
        assert b0 == 0 or n == 1                                # byte ranges only inside one chunk
        for c in range(c0, c0 + n):
            for b in range(b0, b0 + nb):
                seen[c][b] += 1
    assert all(v == 1 for row in seen for v in row)
    if chunks * chunk_bytes <= slot:
        assert _pieces(chunks, chunk_bytes, slot) == [(0, chunks, 0, chunk_bytes)]


def test_partition_carries_the_transport_state():
    """The agreed-decision cache and the channel live on the partition object (a process-global cache
    keyed on id(group) outlives its group)."""
    from sbmc_amd.dist import SlabPartition
    a, b = SlabPartition(720, 8, 3), SlabPartition(720, 8, 3)
    assert a.channel is None and a._agreed == {} and a._agreed is not b._agreed
    assert [SlabPartition(720, 8, r).rows for r in range(8)] == [92, 92, 92, 92, 88, 88, 88, 88]
