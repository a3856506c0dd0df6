"""Delimited RNA table reader (reference: src/ginfinity/table.py:11-117).

Host-side I/O glue in front of the hot path; same signature and error text as
the reference's ``read_rna_table`` so the CLI and user scripts are drop-in.
"""
from __future__ import annotations

import csv
from pathlib import Path
from typing import Iterable, TextIO

from .records import RNA, InputValidationError, _check_column_names


def _resolve_window_columns(fieldnames: Iterable[str], source: str,
                            start_column: str | None,
                            end_column: str | None
                            ) -> tuple[str | None, str | None]:
    """Window columns are optional in the file: both present, or neither."""
    if not (start_column and end_column):
        return start_column, end_column
    have = set(fieldnames)
    absent = [c for c in (start_column, end_column) if c not in have]
    if len(absent) == 2:
        return None, None
    if absent:
        raise ValueError(
            f"RNA table {source} is missing column(s): " + ", ".join(absent))
    return start_column, end_column


def _parse(handle: TextIO, *, source: str, identifier_column: str,
           sequence_column: str, structure_column: str,
           start_column: str | None, end_column: str | None,
           delimiter: str) -> list[RNA]:
    if len(delimiter) != 1:
        raise ValueError("delimiter must be exactly one character")
    if (start_column is None) != (end_column is None):
        raise ValueError("start and end columns must both be provided")
    mandatory = _check_column_names(
        identifier_column, sequence_column, structure_column)
    _check_column_names(identifier_column, sequence_column, structure_column,
                        start_column, end_column)
    reader = csv.DictReader(handle, delimiter=delimiter)
    header = reader.fieldnames
    if header is None:
        raise ValueError(f"empty RNA table: {source}")
    if len(header) != len(set(header)):
        raise ValueError(f"duplicate column name in RNA table: {source}")
    absent = [column for column in mandatory if column not in header]
    if absent:
        raise ValueError(
            f"RNA table {source} is missing column(s): " + ", ".join(absent))
    start_column, end_column = _resolve_window_columns(
        header, source, start_column, end_column)

    seen: set[str] = set()
    records: list[RNA] = []
    for row in reader:
        where = f"RNA table {source} line {reader.line_num}"
        if None in row:
            raise ValueError(f"{where} has extra fields")
        try:
            expanded = RNA.many_from_mapping(
                row, identifier_column=identifier_column,
                sequence_column=sequence_column,
                structure_column=structure_column,
                start_column=start_column, end_column=end_column,
                suffix_identifier=True)
        except InputValidationError as error:
            raise InputValidationError(f"{where}: {error}") from error
        for record in expanded:
            if record.identifier in seen:
                raise InputValidationError(
                    f"{where}: duplicate identifier {record.identifier!r}")
            seen.add(record.identifier)
        records.extend(expanded)
    if not records:
        raise ValueError(f"RNA table contains no records: {source}")
    return records


def read_rna_table(
    path: str | Path,
    *,
    identifier_column: str = "transcript_id",
    sequence_column: str = "sequence",
    structure_column: str = "secondary_structure",
    start_column: str | None = "start",
    end_column: str | None = "end",
    delimiter: str = "\t",
) -> list[RNA]:
    """Read validated RNAs, in input order, from a delimited text table.

    Optional ``start``/``end`` columns may hold one window or parallel
    comma-separated windows per row; each window becomes its own record named
    ``{id}:{start}-{end}``.
    """
    path = Path(path)
    try:
        with path.open(newline="") as handle:
            return _parse(
                handle, source=str(path),
                identifier_column=identifier_column,
                sequence_column=sequence_column,
                structure_column=structure_column,
                start_column=start_column, end_column=end_column,
                delimiter=delimiter)
    except UnicodeDecodeError as error:
        raise ValueError(f"RNA table is not valid text: {path}") from error


__all__ = ["read_rna_table"]
