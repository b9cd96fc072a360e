"""Bit types of the PoT-PTQ scheme (mirror of the reference's models/ptq/bit_type.py:7-57)."""


class BitType:

    def __init__(self, bits, signed, name=None):
        self.bits = bits
        self.signed = signed
        self.name = name if name is not None else ('int' if signed else 'uint') + str(bits)

    @property
    def upper_bound(self):
        return 2**(self.bits - 1) - 1 if self.signed else 2**self.bits - 1

    @property
    def lower_bound(self):
        return -(2**(self.bits - 1)) if self.signed else 0

    @property
    def range(self):
        return 2**self.bits

    def __repr__(self):
        return 'BitType(%s)' % self.name


# order matters: the calibration loop walks this list (models/ptq/layers.py:151-170)
BIT_TYPE_LIST = [BitType(3, False, 'uint3'), BitType(4, False, 'uint4'), BitType(4, True, 'int4'),
                 BitType(8, True, 'int8'), BitType(8, False, 'uint8')]
BIT_TYPE_DICT = {b.name: b for b in BIT_TYPE_LIST}
