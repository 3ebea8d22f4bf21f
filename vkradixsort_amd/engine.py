"""Python spelling of the reference's host interface for the multi_radixsort path, over the C ABI.

The C++ mirror (vkradixsort_amd/host) is the drop-in for C++ callers; this module gives tests and
bench.py the same objects with the same names, argument meaning and error behaviour:

  GPUContext          engine/include/engine/core/GPUContext.h:15-111
  Buffer              engine/include/engine/core/Buffer.h:14-177
  ComputePass         engine/include/engine/passes/ComputePass.h:6-117, Pass.h:54-104
  MultiRadixSortPass  multiradixsort/include/MultiRadixSortPass.h:7-41, src/MultiRadixSortPass.cpp:10-20
  MultiRadixSort      multiradixsort/include/MultiRadixSort.h:9-54, src/MultiRadixSort.cpp:5-161
  SingleRadixSortPass / SingleRadixSort   singleradixsort/...

Vulkan objects have no counterpart: descriptor sets become a (multiBufferedIndex, set, binding) ->
Buffer table, semaphores become opaque tokens (one in-order HIP stream already serialises submits),
the W->R pipeline barriers become stream order.
"""
from __future__ import annotations

import ctypes
import sys
import time
from dataclasses import dataclass

import numpy as np

from . import capi
from .capi import PushConstants, VrsError

RADIX_SORT_BINS = 256
WORKGROUP_SIZE = 256


@dataclass
class Extent3D:
    width: int
    height: int
    depth: int


class GPUContext:
    """Device lifetime + the activeIndex toggle that selects the live descriptor copy."""

    MAX_FRAMES_IN_FLIGHT = 2  # GPUContext.h:110

    def __init__(self, device_ordinal: int = 0, stream: int | None = None):
        self.device_ordinal = device_ordinal
        self._borrowed_stream = stream
        self.m_activeIndex = 0
        self._h = None
        self._lib = None

    def init(self) -> None:  # GPUContext.cpp:7-9
        self._lib = capi.load_library()
        h = ctypes.c_void_p()
        if self._borrowed_stream is None:
            rc = self._lib.vrs_context_create(self.device_ordinal, ctypes.byref(h))
        else:
            rc = self._lib.vrs_context_create_on_stream(self.device_ordinal, ctypes.c_void_p(self._borrowed_stream),
                                                        ctypes.byref(h))
        capi.check(None, rc)
        self._h = h

    def shutdown(self) -> None:  # GPUContext.cpp:11-13
        if self._h is not None:
            self._lib.vrs_context_destroy(self._h)
            self._h = None

    @property
    def handle(self):
        if self._h is None:
            raise RuntimeError("GPUContext is not initialised (call init())")
        return self._h

    @property
    def lib(self):
        return self._lib

    def check(self, rc: int) -> None:
        capi.check(self._h, rc)

    def getMultiBufferedCount(self) -> int:  # GPUContext.h:30-32
        return self.MAX_FRAMES_IN_FLIGHT

    def getActiveIndex(self) -> int:  # GPUContext.h:34-36
        return self.m_activeIndex

    def incrementActiveIndex(self) -> None:  # GPUContext.h:71-73
        self.m_activeIndex = (self.m_activeIndex + 1) % self.MAX_FRAMES_IN_FLIGHT

    def waitIdle(self) -> None:  # vkQueueWaitIdle(COMPUTE), MultiRadixSort.cpp:62
        self.check(self._lib.vrs_queue_wait_idle(self.handle))

    def deviceInfo(self):
        name = ctypes.create_string_buffer(256)
        cus = ctypes.c_int(0)
        mem = ctypes.c_uint64(0)
        self.check(self._lib.vrs_device_info(self.handle, name, 256, ctypes.byref(cus), ctypes.byref(mem)))
        return name.value.decode(), cus.value, mem.value

    # measurement helpers (no reference counterpart)
    def profileEnable(self, on: bool) -> None:
        self.check(self._lib.vrs_profile_enable(self.handle, 1 if on else 0))

    def profileEnableMask(self, kernel_mask: int) -> None:
        self.check(self._lib.vrs_profile_enable_mask(self.handle, kernel_mask))

    def profileReset(self) -> None:
        self.check(self._lib.vrs_profile_reset(self.handle))

    def profileQuery(self, kernel_id: int):
        n = ctypes.c_uint64(0)
        ms = ctypes.c_double(0.0)
        self.check(self._lib.vrs_profile_query(self.handle, kernel_id, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value

    def setTuning(self, key: int, value: int) -> None:
        self.check(self._lib.vrs_set_tuning(self.handle, key, value))

    def __enter__(self):
        self.init()
        return self

    def __exit__(self, *exc):
        self.shutdown()


class Buffer:
    """Device-local buffer; upload/download are synchronous staging copies (Buffer.h:47-72)."""

    @dataclass
    class BufferSettings:  # Buffer.h:16-23 (usage / memory-property flags have no HIP meaning)
        m_sizeBytes: int
        m_name: str = "undefined"

    def __init__(self, gpuContext: GPUContext, settings: "Buffer.BufferSettings", device_ptr: int | None = None):
        self.m_gpuContext = gpuContext
        self.m_bufferSettings = settings
        h = ctypes.c_void_p()
        lib = gpuContext.lib
        if device_ptr is None:
            gpuContext.check(lib.vrs_buffer_create(gpuContext.handle, settings.m_sizeBytes, ctypes.byref(h)))
        else:  # caller-owned device memory ("own usage")
            gpuContext.check(lib.vrs_buffer_wrap(gpuContext.handle, ctypes.c_void_p(device_ptr), settings.m_sizeBytes,
                                                 ctypes.byref(h)))
        self._h = h

    @staticmethod
    def fillDeviceWithStagingBuffer(gpuContext: GPUContext, settings: "Buffer.BufferSettings", data) -> "Buffer":
        buf = Buffer(gpuContext, settings)
        arr = np.ascontiguousarray(data)
        if arr.nbytes < settings.m_sizeBytes:
            # the reference reads m_sizeBytes from `data` unconditionally (Buffer.h:52) -- a host OOB read
            # for tiny N; here it is an error instead
            buf.release()
            raise VrsError(capi.VRS_ERROR_INVALID_ARGUMENT, "host data is smaller than the buffer")
        gpuContext.check(gpuContext.lib.vrs_buffer_upload(gpuContext.handle, buf._h, arr.ctypes.data_as(ctypes.c_void_p),
                                                          settings.m_sizeBytes))
        return buf

    def downloadWithStagingBuffer(self, data: np.ndarray) -> None:
        if not data.flags["C_CONTIGUOUS"] or data.nbytes < self.m_bufferSettings.m_sizeBytes:
            raise VrsError(capi.VRS_ERROR_INVALID_ARGUMENT, "download target must be contiguous and large enough")
        ctx = self.m_gpuContext
        ctx.check(ctx.lib.vrs_buffer_download(ctx.handle, self.handle, data.ctypes.data_as(ctypes.c_void_p),
                                              self.m_bufferSettings.m_sizeBytes))

    def verifyKeys(self, num_elements: int):
        """On-device verify (vrs_verify_keys_u32): (descents, key sum, key mix) of the first num_elements uint32 keys --
        descents == 0 means ascending; sum and mix are order-independent fingerprints of the multiset."""
        ctx = self.m_gpuContext
        d, s, m = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        ctx.check(ctx.lib.vrs_verify_keys_u32(ctx.handle, self.handle, num_elements, ctypes.byref(d), ctypes.byref(s), ctypes.byref(m)))
        return d.value, s.value, m.value

    def copyFrom(self, src: "Buffer", size_bytes: int | None = None) -> None:
        ctx = self.m_gpuContext
        n = self.m_bufferSettings.m_sizeBytes if size_bytes is None else size_bytes
        ctx.check(ctx.lib.vrs_buffer_copy(ctx.handle, self.handle, src.handle, n))

    def release(self) -> None:  # idempotent, Buffer.h:36-45
        if self._h is not None:
            self.m_gpuContext.lib.vrs_buffer_release(self._h)
            self._h = None

    @property
    def handle(self):
        if self._h is None:
            raise VrsError(capi.VRS_ERROR_INVALID_ARGUMENT, "buffer was released")
        return self._h

    def getSizeBytes(self) -> int:  # Buffer.h:103-105
        return self.m_bufferSettings.m_sizeBytes

    def getDeviceAddress(self) -> int:  # Buffer.h:107-110
        return self.m_gpuContext.lib.vrs_buffer_device_ptr(self.handle) or 0

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class ComputePass:
    """Stage bookkeeping of ComputePass/Pass: launch shapes and the binding table."""

    NUM_STAGES = 1

    def __init__(self, gpuContext: GPUContext):
        self.m_gpuContext = gpuContext
        self.m_workGroupCounts: list[Extent3D] = []
        self.m_bindings: list[dict] = []
        self._created = False

    def create(self) -> None:  # ComputePass.h:11-14 / Pass.h:18-32
        self.m_workGroupCounts = [Extent3D(0, 0, 0) for _ in range(self.NUM_STAGES)]
        self.m_bindings = [dict() for _ in range(self.m_gpuContext.getMultiBufferedCount())]
        self._created = True

    def release(self) -> None:  # Pass.h:34-52
        self.m_bindings = []
        self._created = False

    @staticmethod
    def getDispatchSize(width: int, height: int, depth: int, workGroupSize: Extent3D) -> Extent3D:
        # ComputePass.h:24-29
        return Extent3D((width + workGroupSize.width - 1) // workGroupSize.width,
                        (height + workGroupSize.height - 1) // workGroupSize.height,
                        (depth + workGroupSize.depth - 1) // workGroupSize.depth)

    def setGlobalInvocationSize(self, stageIndex: int, width: int, height: int, depth: int) -> None:
        # ComputePass.h:16-22; every shader of this path declares local_size_x = 256
        self.m_workGroupCounts[stageIndex] = self.getDispatchSize(width, height, depth, Extent3D(WORKGROUP_SIZE, 1, 1))

    def getWorkGroupCount(self, stageIndex: int) -> Extent3D:  # ComputePass.h:58-60
        return self.m_workGroupCounts[stageIndex]

    def setStorageBuffer(self, *args) -> None:
        """setStorageBuffer(set, binding, buffer)  -> every descriptor copy   (Pass.h:54-79)
        setStorageBuffer(multiBufferedIndex, set, binding, buffer) -> one copy (Pass.h:81-104)"""
        if len(args) == 3:
            s, b, buf = args
            for copy in self.m_bindings:
                copy[(s, b)] = buf
        elif len(args) == 4:
            idx, s, b, buf = args
            self.m_bindings[idx][(s, b)] = buf
        else:
            raise TypeError("setStorageBuffer takes (set, binding, buffer) or (multiBufferedIndex, set, binding, buffer)")

    def _bound(self, s: int, b: int) -> Buffer:
        try:
            return self.m_bindings[self.m_gpuContext.getActiveIndex()][(s, b)]
        except KeyError:
            raise VrsError(capi.VRS_ERROR_INVALID_ARGUMENT, f"no storage buffer bound at (set {s}, binding {b})") from None

    def execute(self, awaitBeforeExecution=None):  # ComputePass.h:31-56
        if not self._created:
            raise VrsError(capi.VRS_ERROR_INVALID_ARGUMENT, "pass was not created")
        self.recordCommands()
        # the returned "semaphore": stream order already chains submits, the token is only API shape
        return ("signal", self.m_gpuContext.getActiveIndex())

    def recordCommands(self) -> None:
        raise NotImplementedError


class MultiRadixSortPass(ComputePass):
    RADIX_SORT_HISTOGRAMS = 0  # MultiRadixSortPass.h:12-15 (also the descriptor-set numbers)
    RADIX_SORT = 1
    NUM_STAGES = 2

    def __init__(self, gpuContext: GPUContext):
        super().__init__(gpuContext)
        self.m_pushConstantsHistogram = PushConstants()  # MultiRadixSortPass.h:24
        self.m_pushConstants = PushConstants()  # MultiRadixSortPass.h:33
        self.m_pairs = False  # build extension: bindings (1,3)/(1,4) carry values in/out
        self.m_sort64Bit = False  # the reference's SORT_64_BIT switch (MultiRadixSort.h:10-18): uint64 keys

    def recordCommands(self) -> None:  # MultiRadixSortPass.cpp:10-20
        ctx = self.m_gpuContext
        lib = ctx.lib
        hist = lib.vrs_multi_radixsort_histograms_u64 if self.m_sort64Bit else lib.vrs_multi_radixsort_histograms
        sort_pairs = lib.vrs_multi_radixsort_pairs_u64 if self.m_sort64Bit else lib.vrs_multi_radixsort_pairs
        sort_keys = lib.vrs_multi_radixsort_u64 if self.m_sort64Bit else lib.vrs_multi_radixsort
        ctx.check(hist(ctx.handle, self._bound(0, 0).handle, self._bound(0, 1).handle,
                       ctypes.byref(self.m_pushConstantsHistogram)))
        if self.m_pairs:
            ctx.check(sort_pairs(ctx.handle, self._bound(1, 0).handle, self._bound(1, 1).handle,
                                                    self._bound(1, 3).handle, self._bound(1, 4).handle,
                                                    self._bound(1, 2).handle, ctypes.byref(self.m_pushConstants)))
        else:
            ctx.check(sort_keys(ctx.handle, self._bound(1, 0).handle, self._bound(1, 1).handle,
                                self._bound(1, 2).handle, ctypes.byref(self.m_pushConstants)))


class SingleRadixSortPass(ComputePass):
    RADIX_SORT = 0  # SingleRadixSortPass.h:12-14
    NUM_STAGES = 1

    @dataclass
    class PushConstants:  # SingleRadixSortPass.h:16-18
        g_num_elements: int = 0

    def __init__(self, gpuContext: GPUContext):
        super().__init__(gpuContext)
        self.m_pushConstants = SingleRadixSortPass.PushConstants()

    def recordCommands(self) -> None:
        ctx = self.m_gpuContext
        ctx.check(ctx.lib.vrs_single_radixsort(ctx.handle, self._bound(0, 0).handle, self._bound(0, 1).handle,
                                               self.m_pushConstants.g_num_elements))


def generateRandomNumbers(numElements: int, seed: int = 1, reference_28bit: bool = False) -> np.ndarray:
    """std::mt19937(seed)() raw outputs (== numpy RandomState(seed) 32-bit draws).  The reference seeds from
    random_device and keeps only 28 bits (MultiRadixSort.cpp:121-133); `reference_28bit` reproduces that
    range (libstdc++ maps the distribution to raw >> 4)."""
    keys = np.random.RandomState(seed).randint(0, 2 ** 32, size=numElements, dtype=np.uint32)
    return keys >> np.uint32(4) if reference_28bit else keys


class MultiRadixSort:
    """Program logic of multiradixsort/src/MultiRadixSort.cpp with NUM_ELEMENTS as a runtime argument
    (the reference fixes it at compile time, MultiRadixSort.h:29)."""

    PRINT_PREFIX = "[MultiRadixSort] "
    RADIX_SORT_BINS = RADIX_SORT_BINS

    def __init__(self, NUM_ELEMENTS: int = 1000000, NUM_BLOCKS_PER_WORKGROUP: int = 32, seed: int = 1,
                 keys: np.ndarray | None = None, values: np.ndarray | None = None, quiet: bool = False):
        self.NUM_ELEMENTS = int(NUM_ELEMENTS if keys is None else keys.size)
        # SORT_TYPE: uint32 (SORT_32BIT, four passes) unless uint64 keys are handed in (SORT_64_BIT, eight passes)
        self.SORT_TYPE = np.uint64 if (keys is not None and keys.dtype == np.uint64) else np.uint32
        self.KEY_BYTES = np.dtype(self.SORT_TYPE).itemsize
        self.NUM_ELEMENTS_BYTES = self.NUM_ELEMENTS * self.KEY_BYTES
        self.NUM_BLOCKS_PER_WORKGROUP = int(NUM_BLOCKS_PER_WORKGROUP)
        self.seed = seed
        self.m_elementsIn = None if keys is None else np.ascontiguousarray(keys, dtype=self.SORT_TYPE)
        self.m_valuesIn = None if values is None else np.ascontiguousarray(values, dtype=np.uint32)
        self.m_buffers: list = [None, None, None]
        self.m_valueBuffers: list = [None, None]
        self.m_pass: MultiRadixSortPass | None = None
        self.m_gpuContext: GPUContext | None = None
        self.quiet = quiet
        # False (default): enqueueSort() drives the two stages pass by pass like the reference's loop.  True: the
        # library runs the passes itself (vrs_sort_keys_u32 / _u64 / vrs_sort_pairs_u32: one counting read + look-back
        # scatter passes from 2^13 keys on).  Same buffers, same result in buffer 0.  (C++: m_oneCallSort.)
        self.m_oneCallSort = False
        self.gpuSortTime = None
        self.cpuSortTime = None
        self.sorted_keys = None
        self.sorted_values = None

    def _print(self, msg: str) -> None:
        if not self.quiet:
            print(self.PRINT_PREFIX + msg)

    # -- the pieces of execute(), callable on their own by bench.py --------------------------------------
    def setup(self, gpuContext: GPUContext) -> None:  # MultiRadixSort.cpp:6-46
        self.m_gpuContext = gpuContext
        self.m_pass = MultiRadixSortPass(gpuContext)
        self.m_pass.create()
        self.m_pass.m_sort64Bit = self.KEY_BYTES == 8
        B = self.NUM_BLOCKS_PER_WORKGROUP
        globalInvocationSize = self.NUM_ELEMENTS // B
        remainder = self.NUM_ELEMENTS % B
        globalInvocationSize += 1 if remainder > 0 else 0
        self.m_pass.setGlobalInvocationSize(MultiRadixSortPass.RADIX_SORT_HISTOGRAMS, globalInvocationSize, 1, 1)
        self.m_pass.setGlobalInvocationSize(MultiRadixSortPass.RADIX_SORT, globalInvocationSize, 1, 1)
        NUM_WORKGROUPS = self.m_pass.getWorkGroupCount(MultiRadixSortPass.RADIX_SORT_HISTOGRAMS).width
        assert NUM_WORKGROUPS == self.m_pass.getWorkGroupCount(MultiRadixSortPass.RADIX_SORT).width
        for pc in (self.m_pass.m_pushConstantsHistogram, self.m_pass.m_pushConstants):
            pc.g_num_elements = self.NUM_ELEMENTS
            pc.g_num_workgroups = NUM_WORKGROUPS
            pc.g_num_blocks_per_workgroup = B
        self.prepareBuffers()
        self.bindBuffers()

    def prepareBuffers(self) -> None:  # MultiRadixSort.cpp:83-95
        ctx = self.m_gpuContext
        if self.m_elementsIn is None:
            self.m_elementsIn = generateRandomNumbers(self.NUM_ELEMENTS, self.seed)
        W = self.m_pass.getWorkGroupCount(MultiRadixSortPass.RADIX_SORT_HISTOGRAMS).width
        S = Buffer.BufferSettings
        self.m_buffers[0] = Buffer.fillDeviceWithStagingBuffer(ctx, S(self.NUM_ELEMENTS_BYTES, "radixSort.elementBuffer0"),
                                                               self.m_elementsIn)
        self.m_buffers[1] = Buffer(ctx, S(self.NUM_ELEMENTS_BYTES, "radixSort.elementBuffer1"))
        # the reference zero-fills this buffer from a too-short host vector (MultiRadixSort.cpp:93-94); the
        # histogram stage overwrites every entry, so no initialisation is needed
        self.m_buffers[2] = Buffer(ctx, S(W * RADIX_SORT_BINS * 4, "radixSort.histogramsBuffer"))
        if self.m_valuesIn is not None:
            self.m_pass.m_pairs = True
            self.m_valueBuffers[0] = Buffer.fillDeviceWithStagingBuffer(ctx, S(self.NUM_ELEMENTS * 4, "radixSort.valueBuffer0"),
                                                                        self.m_valuesIn)
            self.m_valueBuffers[1] = Buffer(ctx, S(self.NUM_ELEMENTS * 4, "radixSort.valueBuffer1"))

    def bindBuffers(self) -> None:  # MultiRadixSort.cpp:33-46
        p = self.m_pass
        a = self.m_gpuContext.getActiveIndex()
        o = (a + 1) % 2
        H, R = MultiRadixSortPass.RADIX_SORT_HISTOGRAMS, MultiRadixSortPass.RADIX_SORT
        p.setStorageBuffer(a, H, 0, self.m_buffers[0])  # iteration 0 and 2 (0,0)
        p.setStorageBuffer(a, R, 0, self.m_buffers[0])  # iteration 0 and 2 (1,0)
        p.setStorageBuffer(o, R, 1, self.m_buffers[0])  # iteration 1 and 3 (1,1)
        p.setStorageBuffer(o, H, 0, self.m_buffers[1])  # iteration 1 and 3 (0,0)
        p.setStorageBuffer(a, R, 1, self.m_buffers[1])  # iteration 0 and 2 (1,1)
        p.setStorageBuffer(o, R, 0, self.m_buffers[1])  # iteration 1 and 3 (1,0)
        p.setStorageBuffer(H, 1, self.m_buffers[2])
        p.setStorageBuffer(R, 2, self.m_buffers[2])
        if p.m_pairs:
            p.setStorageBuffer(a, R, 3, self.m_valueBuffers[0])
            p.setStorageBuffer(a, R, 4, self.m_valueBuffers[1])
            p.setStorageBuffer(o, R, 3, self.m_valueBuffers[1])
            p.setStorageBuffer(o, R, 4, self.m_valueBuffers[0])

    def enqueueSort(self) -> None:  # the hot loop, MultiRadixSort.cpp:50-61 (no blocking call inside)
        ctx = self.m_gpuContext
        if self.m_oneCallSort:
            k0, k1 = self.m_buffers[0].handle, self.m_buffers[1].handle
            if self.m_pass.m_pairs:
                fn = ctx.lib.vrs_sort_pairs_u64 if self.KEY_BYTES == 8 else ctx.lib.vrs_sort_pairs_u32
                ctx.check(fn(ctx.handle, k0, k1, self.m_valueBuffers[0].handle, self.m_valueBuffers[1].handle,
                             self.NUM_ELEMENTS))
            elif self.KEY_BYTES == 8:
                ctx.check(ctx.lib.vrs_sort_keys_u64(ctx.handle, k0, k1, self.NUM_ELEMENTS))
            else:
                ctx.check(ctx.lib.vrs_sort_keys_u32(ctx.handle, k0, k1, self.NUM_ELEMENTS))
            return
        awaitBeforeExecution = None
        NUM_ITERATIONS = self.KEY_BYTES  # 4 (SORT_32BIT) or 8 (SORT_64_BIT), MultiRadixSort.cpp:50-55
        for i in range(NUM_ITERATIONS):
            self.m_pass.m_pushConstantsHistogram.g_shift = 8 * i
            self.m_pass.m_pushConstants.g_shift = 8 * i
            awaitBeforeExecution = self.m_pass.execute(awaitBeforeExecution)
            self.m_gpuContext.incrementActiveIndex()

    @staticmethod
    def sort(buffer: np.ndarray) -> float:  # MultiRadixSort.cpp:141-146 (np.sort stands in for std::sort)
        begin = time.perf_counter()
        buffer.sort(kind="quicksort")
        return (time.perf_counter() - begin) * 1e3

    def testSort(self, reference: np.ndarray, outBuffer: np.ndarray) -> bool:  # MultiRadixSort.cpp:148-161
        if reference.size != outBuffer.size:
            print(self.PRINT_PREFIX + "reference.size() != outBuffer.size()", file=sys.stderr)
            raise RuntimeError("TEST FAILED.")
        neq = np.flatnonzero(reference != outBuffer)
        if neq.size:
            i = int(neq[0])
            print(f"{self.PRINT_PREFIX}{reference[i]} = reference[{i}] != outBuffer[{i}] = {outBuffer[i]}", file=sys.stderr)
            raise RuntimeError("TEST FAILED.")
        self._print("Test passed.")
        return True

    def download(self) -> np.ndarray:  # verify()'s download, MultiRadixSort.cpp:97-100: result is in buffer0
        data = np.empty(self.NUM_ELEMENTS, dtype=self.SORT_TYPE)
        self.m_buffers[0].downloadWithStagingBuffer(data)
        return data

    def releaseBuffers(self) -> None:  # MultiRadixSort.cpp:115-119
        for b in self.m_buffers + self.m_valueBuffers:
            if b is not None:
                b.release()

    def execute(self, gpuContext: GPUContext) -> None:  # MultiRadixSort.cpp:5-81
        self.setup(gpuContext)
        self._print(f"Sorting {self.NUM_ELEMENTS} {8 * self.KEY_BYTES}bit numbers.")
        begin = time.perf_counter()
        self.enqueueSort()
        gpuContext.waitIdle()
        self.gpuSortTime = (time.perf_counter() - begin) * 1e3
        self._print(f"GPU sort finished in {self.gpuSortTime:.3f}[ms].")
        self.sorted_keys = self.download()
        if self.m_valuesIn is not None:
            self.sorted_values = np.empty(self.NUM_ELEMENTS, dtype=np.uint32)
            self.m_valueBuffers[0].downloadWithStagingBuffer(self.sorted_values)
            order = np.argsort(self.m_elementsIn, kind="stable")
            ref_keys, ref_vals = self.m_elementsIn[order], self.m_valuesIn[order]
            self.cpuSortTime = 0.0
            self.testSort(ref_keys, self.sorted_keys)
            self.testSort(ref_vals, self.sorted_values)
        else:
            reference = self.m_elementsIn.copy()
            self.cpuSortTime = self.sort(reference)
            self._print(f"CPU sort finished in {self.cpuSortTime:.3f}[ms].")
            self.testSort(reference, self.sorted_keys)
        self.releaseBuffers()
        self.m_pass.release()


class SingleRadixSort:
    """singleradixsort/src/SingleRadixSort.cpp:5-47 with NUM_ELEMENTS as a runtime argument."""

    PRINT_PREFIX = "[SingleRadixSort] "
    INPUT_BUFFER_INDEX = 0

    def __init__(self, NUM_ELEMENTS: int = 1000000, seed: int = 1, keys: np.ndarray | None = None, quiet: bool = False):
        self.NUM_ELEMENTS = int(NUM_ELEMENTS if keys is None else keys.size)
        self.NUM_ELEMENTS_BYTES = self.NUM_ELEMENTS * 4
        self.m_elementsIn = generateRandomNumbers(self.NUM_ELEMENTS, seed) if keys is None else np.ascontiguousarray(keys, np.uint32)
        self.quiet = quiet
        self.sorted_keys = None

    def execute(self, gpuContext: GPUContext) -> None:
        p = SingleRadixSortPass(gpuContext)
        p.create()
        p.setGlobalInvocationSize(SingleRadixSortPass.RADIX_SORT, 256, 1, 1)  # exactly one workgroup (:12)
        p.m_pushConstants.g_num_elements = self.NUM_ELEMENTS
        S = Buffer.BufferSettings
        b0 = Buffer.fillDeviceWithStagingBuffer(gpuContext, S(self.NUM_ELEMENTS_BYTES, "radixSort.elementBuffer0"), self.m_elementsIn)
        b1 = Buffer(gpuContext, S(self.NUM_ELEMENTS_BYTES, "radixSort.elementBuffer1"))
        p.setStorageBuffer(SingleRadixSortPass.RADIX_SORT, 0, b0)
        p.setStorageBuffer(SingleRadixSortPass.RADIX_SORT, 1, b1)
        if not self.quiet:
            print(f"{self.PRINT_PREFIX}Sorting {self.NUM_ELEMENTS} 32bit numbers.")
        begin = time.perf_counter()
        p.execute(None)
        gpuContext.waitIdle()
        self.gpuSortTime = (time.perf_counter() - begin) * 1e3
        data = np.empty(self.NUM_ELEMENTS, dtype=np.uint32)
        b0.downloadWithStagingBuffer(data)
        self.sorted_keys = data
        reference = np.sort(self.m_elementsIn)
        if reference.size != data.size or np.any(reference != data):
            raise RuntimeError("TEST FAILED.")
        if not self.quiet:
            print(f"{self.PRINT_PREFIX}Test passed.")
        b0.release()
        b1.release()
        p.release()
