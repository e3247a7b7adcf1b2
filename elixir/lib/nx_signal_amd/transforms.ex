defmodule NxSignalAMD.Transforms do
  @moduledoc """
  `NxSignal.Transforms.fft_nd/2` and `ifft_nd/2` (lib/nx_signal/transforms.ex:5-21): `Nx.fft` / `Nx.ifft` folded over
  `axes:` with `lengths:` — the whole fold runs in HBM (tiled transposes between axes, four-step rows beyond 8192 points,
  Bluestein rows for non-power-of-two lengths).
  """
  alias NxSignalAMD.NIF

  def fft_nd(tensor, opts \\ []), do: run(tensor, opts, 0)
  def ifft_nd(tensor, opts \\ []), do: run(tensor, opts, 1)

  defp run(%Nx.Tensor{} = tensor, opts, inverse) do
    opts = Keyword.validate!(opts, axes: [-1], lengths: nil)
    axes = opts[:axes]
    shape = Nx.shape(tensor) |> Tuple.to_list()
    rank = length(shape)
    axes_n = Enum.map(axes, fn ax -> if ax < 0, do: ax + rank, else: ax end)
    lengths = (opts[:lengths] || List.duplicate(nil, length(axes))) |> Enum.zip(axes_n) |> Enum.map(fn {l, ax} -> l || Enum.at(shape, ax) end)

    if Nx.type(tensor) in [{:f, 64}, {:c, 128}] do
      run_f64(tensor, shape, rank, axes_n, lengths, inverse)
    else
      run_f32(tensor, shape, axes_n, lengths, inverse)
    end
  end

  # f64 / c128 tensors: Nx.fft / Nx.ifft in c128 over the LAST axis (the f64 tier, include/nxsig.h); other axes are not built in double
  defp run_f64(tensor, shape, rank, axes_n, lengths, inverse) do
    if axes_n != [rank - 1] do
      raise ArgumentError, "fft_nd: f64 / c128 tensors are transformed over the last axis only"
    end

    [k] = lengths
    n_in = List.last(shape)
    rows = div(Enum.product(shape), n_in)
    is_real = if Nx.type(tensor) == {:f, 64}, do: 1, else: 0

    {:ok, out} =
      NIF.fft_c128(NxSignalAMD.context(), Nx.to_binary(tensor), is_real, rows, n_in, k, inverse) |> NxSignalAMD.unwrap!()

    Nx.from_binary(out, :c128) |> Nx.reshape(shape |> List.replace_at(rank - 1, k) |> List.to_tuple())
  end

  defp run_f32(tensor, shape, axes_n, lengths, inverse) do
    {bin, is_real} =
      case Nx.type(tensor) do
        {:c, 64} -> {Nx.to_binary(tensor), 0}
        _ -> {tensor |> Nx.as_type(:f32) |> Nx.to_binary(), 1}
      end

    {:ok, out} = NIF.fft_nd(NxSignalAMD.context(), bin, is_real, shape, axes_n, lengths, inverse) |> NxSignalAMD.unwrap!()

    out_shape =
      Enum.zip(axes_n, lengths) |> Enum.reduce(shape, fn {ax, l}, acc -> List.replace_at(acc, ax, l) end) |> List.to_tuple()

    Nx.from_binary(out, :c64) |> Nx.reshape(out_shape)
  end
end
