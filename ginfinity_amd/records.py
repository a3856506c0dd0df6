"""RNA input records for the MI355X encoder.

Host-side mirror of the reference's input contract
(reference: src/ginfinity/_validation.py:9-258).  The same names, argument
meaning and error messages are kept so that code written against the
reference's ``RNA`` keeps working; the implementation is independent (a single
regex/translate based normaliser and a depth-counter bracket check instead of
the reference's stack walk).
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from numbers import Integral
from typing import Mapping, Sequence

MAXIMUM_LENGTH_NT = 4096          # reference: _validation.py:225, data/model.json:9

_BASES = frozenset("ACGU")
_BRACKETS = frozenset(".()")
_FORBIDDEN_ID_CHARS = re.compile(r"[\t\r\n]")


class InputValidationError(ValueError):
    """An RNA record violates the supported input contract."""


# --------------------------------------------------------------------------
# window parsing (reference: _validation.py:19-64)
# --------------------------------------------------------------------------

def parse_position_list(value: object, *, name: str) -> list[int]:
    """``None``/blank → [], an integer → [n], ``"a, b"`` → [a, b]."""
    if value is None:
        return []
    if isinstance(value, bool):
        raise InputValidationError(f"{name} must be an integer")
    if isinstance(value, Integral):
        return [int(value)]
    if isinstance(value, float):
        if value.is_integer():
            return [int(value)]
        raise InputValidationError(f"{name} must be an integer or integer list")
    if not isinstance(value, str):
        raise InputValidationError(f"{name} must be an integer or integer list")
    stripped = value.strip()
    if not stripped:
        return []
    out: list[int] = []
    for token in stripped.split(","):
        token = token.strip()
        if not token:
            raise InputValidationError(f"empty value in {name} list")
        try:
            out.append(int(token, 10))
        except ValueError as error:
            raise InputValidationError(
                f"invalid integer {token!r} in {name}") from error
    return out


def parse_slice_bounds(start: object, end: object) -> list[tuple[int, int]]:
    """Pair parallel start/end lists into 0-based half-open windows."""
    starts = parse_position_list(start, name="start")
    ends = parse_position_list(end, name="end")
    if len(starts) != len(ends):
        raise InputValidationError(
            f"start has {len(starts)} value(s) but end has {len(ends)}")
    return list(zip(starts, ends))


def sliced_identifier(identifier: str, start: int, end: int) -> str:
    return f"{identifier}:{start}-{end}"


def _check_column_names(*columns: str | None) -> tuple[str, ...]:
    named = tuple(column for column in columns if column)
    if len(set(named)) != len(named):
        raise ValueError("RNA column names must differ")
    return named


# for table.py (reference name kept: _validation.py:82-88)
_validate_column_names = _check_column_names


# --------------------------------------------------------------------------
# sequence / structure normalisation (reference: _validation.py:224-258)
# --------------------------------------------------------------------------

def validate_and_normalize(sequence: str, structure: str, *,
                           maximum_length: int = MAXIMUM_LENGTH_NT
                           ) -> tuple[str, str]:
    sequence = sequence.strip().upper().replace("T", "U")
    structure = structure.strip()
    if not sequence:
        raise InputValidationError("empty sequence")
    if len(sequence) > maximum_length:
        raise InputValidationError(
            f"sequence length {len(sequence)} exceeds maximum {maximum_length}")
    if len(structure) != len(sequence):
        raise InputValidationError(
            f"structure is {len(structure)} characters against a "
            f"{len(sequence)} nt sequence")
    bad = set(sequence) - _BASES
    if bad:
        raise InputValidationError(
            "unsupported sequence character(s): " + " ".join(sorted(bad)))
    bad = set(structure) - _BRACKETS
    if bad:
        raise InputValidationError(
            "unsupported structure character(s): " + " ".join(sorted(bad)))
    # balanced-bracket check: running depth; remember the still-open positions
    # only so the error can name the first unmatched '('.
    open_positions: list[int] = []
    for position, char in enumerate(structure):
        if char == "(":
            open_positions.append(position)
        elif char == ")":
            if not open_positions:
                raise InputValidationError(
                    f"unmatched ')' at 0-based position {position}")
            open_positions.pop()
    if open_positions:
        raise InputValidationError(
            f"unmatched '(' at 0-based position {open_positions[0]}")
    return sequence, structure


# --------------------------------------------------------------------------
# the record type (reference: _validation.py:91-221)
# --------------------------------------------------------------------------

@dataclass(frozen=True, slots=True)
class RNA:
    """One RNA molecule, its dot-bracket structure and an optional window.

    ``start``/``end`` are 0-based half-open coordinates into the normalised
    sequence (``sequence[start:end]``); omit both for a full molecule.
    """

    identifier: str
    sequence: str
    structure: str
    start: int | None = None
    end: int | None = None

    def __post_init__(self) -> None:
        identifier = self.identifier.strip()
        sequence, structure = validate_and_normalize(
            self.sequence, self.structure)
        if not identifier:
            raise InputValidationError("empty identifier")
        if _FORBIDDEN_ID_CHARS.search(identifier):
            raise InputValidationError(
                "identifier must not contain tabs or line breaks")
        start, end = self.start, self.end
        if (start is None) != (end is None):
            raise InputValidationError("start and end must both be provided")
        if start is not None:
            for label, value in (("start", start), ("end", end)):
                if isinstance(value, bool) or not isinstance(value, Integral):
                    raise InputValidationError(f"{label} must be an integer")
            start, end = int(start), int(end)
            if not (0 <= start < end <= len(sequence)):
                raise InputValidationError(
                    f"invalid slice [{start}, {end}) for a "
                    f"{len(sequence)} nt sequence")
        for name, value in (("identifier", identifier), ("sequence", sequence),
                            ("structure", structure), ("start", start),
                            ("end", end)):
            object.__setattr__(self, name, value)

    @property
    def length(self) -> int:
        return len(self.sequence)

    @property
    def sliced(self) -> bool:
        return self.start is not None

    @property
    def core_length(self) -> int:
        return self.length if self.start is None else self.end - self.start

    @classmethod
    def many_from_mapping(
        cls,
        row: Mapping[str, object],
        *,
        identifier_column: str = "transcript_id",
        sequence_column: str = "sequence",
        structure_column: str = "secondary_structure",
        start_column: str | None = "start",
        end_column: str | None = "end",
        suffix_identifier: bool = True,
    ) -> list["RNA"]:
        """One record per window listed in ``row`` (or one full molecule)."""
        if (start_column is None) != (end_column is None):
            raise ValueError("start and end columns must both be provided")
        _check_column_names(identifier_column, sequence_column,
                            structure_column, start_column, end_column)
        wanted: Sequence[str] = (identifier_column, sequence_column,
                                 structure_column)
        absent = [column for column in wanted if column not in row]
        if absent:
            raise InputValidationError(
                "missing RNA column(s): " + ", ".join(absent))
        identifier, sequence, structure = (row[column] for column in wanted)
        if not (isinstance(identifier, str) and isinstance(sequence, str)
                and isinstance(structure, str)):
            raise InputValidationError(
                "RNA identifier, sequence, and structure must be strings")
        windows: list[tuple[int, int]] = []
        if start_column is not None and (start_column in row
                                         or end_column in row):
            absent = [column for column in (start_column, end_column)
                      if column not in row]
            if absent:
                raise InputValidationError(
                    "missing RNA column(s): " + ", ".join(absent))
            windows = parse_slice_bounds(row[start_column], row[end_column])
        if not windows:
            return [cls(identifier, sequence, structure)]
        rename = suffix_identifier or len(windows) > 1
        return [
            cls(sliced_identifier(identifier, lo, hi) if rename else identifier,
                sequence, structure, start=lo, end=hi)
            for lo, hi in windows
        ]

    @classmethod
    def from_mapping(
        cls,
        row: Mapping[str, object],
        *,
        identifier_column: str = "transcript_id",
        sequence_column: str = "sequence",
        structure_column: str = "secondary_structure",
        start_column: str | None = None,
        end_column: str | None = None,
    ) -> "RNA":
        records = cls.many_from_mapping(
            row, identifier_column=identifier_column,
            sequence_column=sequence_column,
            structure_column=structure_column,
            start_column=start_column, end_column=end_column,
            suffix_identifier=False)
        if len(records) != 1:
            raise InputValidationError(
                "mapping defines multiple slices; use RNA.many_from_mapping()")
        return records[0]


__all__ = ["InputValidationError", "RNA", "MAXIMUM_LENGTH_NT",
           "parse_position_list", "parse_slice_bounds", "sliced_identifier",
           "validate_and_normalize"]
